#!/bin/bash
# dev sweep (round 4): the tile shapes of scripts/sweep_c3.sh at the configs[3] shapes (416x416, batch 64) -- the check
# that the cost model of haloq_tile_choice keeps the measured-best kernels there -- and the ring / K-step forms of the
# weight gradient at 208 / 52 / 26 beside the 224x224 findings (ring at 28x28, 128-pixel K steps at 112x112)
cd "$(dirname "$0")/.." || exit 1
export Y2_DEV_LIB=1 Y2DEV_BENCH_ROT=3 BATCH=64
HQ=100,119,126,121,133,123,134
echo "== 3x3 forward (with BN statistics), batch 64"
Y2DEV_BENCH_STATS=1 SHAPES="52,128,256,3;26,256,512,3;13,512,1024,3;13,1024,1024,3" python3 scripts/bench_conv.py $HQ 2>&1 | grep -v amdgpu
echo "== 3x3 dgrad shapes, batch 64"
SHAPES="52,256,128,3;26,512,256,3;13,1024,512,3" python3 scripts/bench_conv.py $HQ 2>&1 | grep -v amdgpu
echo "== weight gradients, batch 64"
SHAPES="208,32,64,3;104,64,128,3;52,128,256,3;26,256,512,3" python3 scripts/bench_wgrad.py 1:0,44:0,46:0,31:0,45:0,47:0,50:0,11:0 2>&1 | grep -v amdgpu
echo "== weight gradients, batch 128 (configs[2]): split depth of the 112x112 ring form"
BATCH=128 SHAPES="112,32,64,3" python3 scripts/bench_wgrad.py 44:0,46:0,44:384,44:256,44:192,44:128 2>&1 | grep -v amdgpu
