#!/bin/bash
# dev sweep: XCD-aware workgroup order on the weight-gradient kernels (f16, N=64)
export Y2_DEV_LIB=1
for x in 0 1; do
echo "Y2_XCD_WGRAD=$x"
Y2_XCD_WGRAD=$x SHAPES="52,256,128,1;26,512,256,1;13,1024,512,1;26,256,512,3;13,512,1024,3;13,1024,1024,3;104,64,128,3;52,128,256,3" python3 scripts/bench_wgrad.py 0:0,1:0
done
