#!/bin/bash
# dev sweep: LDS stages of the nine-tap weight gradient (f16, N=64): 11 = 64x32 two stages, 14 = three stages, 50 = 128-pixel K steps
export Y2_DEV_LIB=1
SHAPES="26,256,512,3;13,512,1024,3;13,1024,1024,3" python3 scripts/bench_wgrad.py 1:0,11:0,14:0,50:0
