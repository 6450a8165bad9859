#!/bin/bash
# dev sweep: ring-form weight gradient of the 208x208 layer (f16, N=64): K-step size and workgroup count
export Y2_DEV_LIB=1
SHAPES="208,32,64,3" python3 scripts/bench_wgrad.py 1:0,31:0,35:0,38:0,44:0,46:0,41:0,40:0
