#!/bin/bash
# The f16x2 part of scripts/gpu_profile_r05.sh on its own (round 5, after the two-plane K loop of the convolution kernels):
# kernel trace + stats + timeline of the split-operand step, its per-layer table, one SQ counter pass.
TAG=${1:-r05f}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=2; K=3; ALL=$((W + K))
X2="python3 bench.py --dtype f16x2 --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --no-extra-legs --kernel-events off --sustain-steps 0 --fed-steps 0"
O=gpurun_out
rm -rf $O/${TAG}_trace_f16x2 $O/${TAG}_sq
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_f16x2 -o run --output-format csv -- $X2 > $O/${TAG}_trace_f16x2.log 2>&1
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_f16x2 $O/${TAG}_stats_f16x2.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_f16x2 $O/${TAG}_timeline_f16x2.csv $ALL
DTYPE=f16x2 python3 scripts/profile_layers.py > $O/${TAG}_layers_f16x2.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_sq -o run --output-format csv -- $X2 > $O/${TAG}_sq_f16x2.log 2>&1
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds_f16x2.csv
rm -rf $O/${TAG}_sq
find $O/${TAG}_trace_f16x2 -type f ! -name "*kernel_stats.csv" -delete
ls -la $O | grep ${TAG}
