#!/bin/bash
# dev sweep: workgroup count of the per-tap (1x1) weight gradient with slab partials (f16, N=64)
export Y2_DEV_LIB=1
SHAPES="52,256,128,1" python3 scripts/bench_wgrad.py 0:0,0:128,0:192,0:256,0:384
SHAPES="26,512,256,1" python3 scripts/bench_wgrad.py 0:0,0:32,0:48,0:64,0:96
SHAPES="13,1024,512,1" python3 scripts/bench_wgrad.py 0:0,0:8,0:12,0:16,0:24
SHAPES="13,1024,32,1" python3 scripts/bench_wgrad.py 0:0,0:8,0:16,0:32,0:64
