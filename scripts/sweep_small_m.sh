#!/bin/bash
# dev sweep (round 4): tile shapes of conv_haloq where a launch has only a few hundred to a few thousand pixels: the
# reference's own training shape (224x224, batch 24: src/pascal/pascal_train_darknet.py:26) and single-image detection
cd "$(dirname "$0")/.." || exit 1
export Y2_DEV_LIB=1 Y2DEV_BENCH_ROT=3
V=100,124,149,147,146,148,130,134,121,123,137
for B in 24 4 1; do
  echo "== batch $B: 3x3 forward shapes (BN statistics) then dgrad shapes"
  BATCH=$B Y2DEV_BENCH_STATS=1 SHAPES="7,512,1024,3;7,1024,1024,3;14,256,512,3;28,128,256,3" python3 scripts/bench_conv.py $V 2>&1 | grep -v amdgpu
  BATCH=$B SHAPES="7,1024,512,3;14,512,256,3;28,256,128,3" python3 scripts/bench_conv.py $V 2>&1 | grep -v amdgpu
done
