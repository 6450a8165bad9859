#!/bin/bash
# dev sweep (round 4): tile shapes of the 3x3 / 1x1 convolutions and the weight gradients at the configs[2] shapes
# (224x224, batch 128: 112 / 56 / 28 / 14 / 7 maps) -- the launch policy was tuned on the 416x416 batch-64 shapes only
# and C3 runs its 14x14 / 7x7 layers at half the rate of their 26x26 / 13x13 siblings (profiles/r04_layers_c3*.txt)
cd "$(dirname "$0")/.." || exit 1
export Y2_DEV_LIB=1 Y2DEV_BENCH_ROT=3 BATCH=${BATCH:-128}
HQ=100,119,120,121,118,123,124,125,130,131,132,133,134,126
echo "== 3x3 forward (with BN statistics)"
Y2DEV_BENCH_STATS=1 SHAPES="${FWD:-28,128,256,3;14,256,512,3;7,512,1024,3}" python3 scripts/bench_conv.py $HQ 2>&1 | grep -v amdgpu
echo "== 3x3 dgrad shapes"
SHAPES="${DGR:-28,256,128,3;14,512,256,3;7,1024,512,3}" python3 scripts/bench_conv.py $HQ 2>&1 | grep -v amdgpu
echo "== 1x1 forward / dgrad shapes"
SHAPES="${ONE:-28,256,128,1;14,512,256,1;7,1024,512,1;7,1024,1000,1;28,128,256,1;14,256,512,1;7,512,1024,1}" python3 scripts/bench_conv.py 100,0,1,2,3,4,6,7,8,207,208 2>&1 | grep -v amdgpu
echo "== weight gradients"
SHAPES="${WGR:-112,32,64,3;56,64,128,3;56,128,64,3;28,128,256,3;14,256,512,3;7,512,1024,3}" python3 scripts/bench_wgrad.py 1:0,11:0,50:0,9:0,51:0,13:0,52:0,45:0,44:0,30:0,31:0,33:0,47:0,46:0 2>&1 | grep -v amdgpu
