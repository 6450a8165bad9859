#!/bin/bash
# dev sweep: nine-tap wgrad with two co sub-tiles per wave vs the product variants (f16, N=64)
export Y2_DEV_LIB=1
SHAPES="26,256,512,3;13,512,1024,3;13,1024,1024,3" python3 scripts/bench_wgrad.py 1:0,50:0,11:0,60:0,61:0,62:0,63:0,64:0,65:0,66:0,68:0,69:0
