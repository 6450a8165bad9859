#!/bin/bash
# dev sweep (round 4): (a) TC = 2 tile forms of conv_haloq on the 64-cout layers of the 104x104 maps (one LDS pixel
# fragment per TWO MFMAs instead of one), (b) the nine-tap weight gradient with 32 x 32 tiles and a shallower split
# (less slab traffic) on the split-K 13x13 / 26x26 layers
cd "$(dirname "$0")/.." || exit 1
export Y2_DEV_LIB=1 Y2DEV_BENCH_ROT=3 BATCH=64
echo "== 104x104 128->64 forward (BN statistics) and the 64-cout dgrad shape"
Y2DEV_BENCH_STATS=1 SHAPES="104,128,64,3" python3 scripts/bench_conv.py 100,133,140,141,142,143,144,145,135 2>&1 | grep -v amdgpu
SHAPES="104,128,64,3;208,64,32,3" python3 scripts/bench_conv.py 100,133,140,141,142,143,144,145,135 2>&1 | grep -v amdgpu
echo "== weight gradients: tile / split"
SHAPES="13,512,1024,3;26,256,512,3;13,1024,1024,3" python3 scripts/bench_wgrad.py 1:0,12:1,12:2,12:0,11:1,11:2,13:1,13:2,16:0,16:1 2>&1 | grep -v amdgpu
