#!/bin/bash
# Round-5 profile recipe (run through gpurun from the repo root).  As gpu_profile_r04.sh for the headline (C4) command --
# kernel trace + stats with the production streams and serialised, SQ / FETCH / WRITE counter passes, per-layer tables --
# plus the split-operand mode (--dtype f16x2: trace + stats + per-layer table), the exact-f32 mode's per-layer table, and
# the ResNet swap's kernel table of one applied, graph-replayed step (batch 32).
# rocprofv3 rule of this pool: the program goes directly after `--`; --pmc passes carry no other trace domain.
TAG=${1:-r05e}
PMC=${2:-1}          # 0: skip the counter passes
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=2; K=3; ALL=$((W + K))
B="python3 bench.py --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --no-fast-parity-mode --no-extra-legs --kernel-events off --sustain-steps 0 --fed-steps 0"
X2="python3 bench.py --dtype f16x2 --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --no-extra-legs --kernel-events off --sustain-steps 0 --fed-steps 0"
C3="python3 bench.py --model classifier"
C2="python3 bench.py --forward-only --batch 32 --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --kernel-events off --sustain-steps 0"
RN="python3 bench.py --model resnet50 --batch 32 --graph --steps 6 --warmup 4 --no-cpu-baseline"
O=gpurun_out
rm -rf $O/${TAG}_trace_overlap $O/${TAG}_trace_serial $O/${TAG}_trace_f16x2 $O/${TAG}_trace_c3 $O/${TAG}_trace_c2 $O/${TAG}_trace_rn $O/${TAG}_sq $O/${TAG}_sq2 $O/${TAG}_fetch $O/${TAG}_write
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_overlap -o run --output-format csv -- $B > $O/${TAG}_trace_overlap.log 2>&1
export Y2_NO_WGRAD_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_serial -o run --output-format csv -- $B > $O/${TAG}_trace_serial.log 2>&1
unset Y2_NO_WGRAD_OVERLAP
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_f16x2 -o run --output-format csv -- $X2 > $O/${TAG}_trace_f16x2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_c3 -o run --output-format csv -- $C3 > $O/${TAG}_trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_c2 -o run --output-format csv -- $C2 > $O/${TAG}_trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_rn -o run --output-format csv -- $RN > $O/${TAG}_trace_rn.log 2>&1
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_overlap $O/${TAG}_stats_overlap.csv $ALL
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_serial $O/${TAG}_stats_serial.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_overlap $O/${TAG}_timeline_overlap.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_serial $O/${TAG}_timeline_serial.csv $ALL
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_f16x2 $O/${TAG}_stats_f16x2.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_f16x2 $O/${TAG}_timeline_f16x2.csv $ALL
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_c3 $O/${TAG}_stats_c3.csv 23        # 3 warm-up + 20 timed steps
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_c2 $O/${TAG}_stats_c2_forward.csv $ALL
# one APPLIED graph-replayed step (bench settles the loss scale first): the dispatches between the last two root convolutions
python3 scripts/summarize_profiles.py step $O/${TAG}_trace_rn $O/${TAG}_resnet50_step.csv rn_conv7_mfma_kernel
python3 scripts/profile_layers.py > $O/${TAG}_layers_per_layer_us.txt 2>&1
DTYPE=f16x2 python3 scripts/profile_layers.py > $O/${TAG}_layers_f16x2.txt 2>&1
DTYPE=f32 python3 scripts/profile_layers.py > $O/${TAG}_layers_f32.txt 2>&1
MODEL=classifier python3 scripts/profile_layers.py > $O/${TAG}_layers_c3.txt 2>&1
if [ "$PMC" = "1" ]; then
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_sq -o run --output-format csv -- $B > $O/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/${TAG}_sq2 -o run --output-format csv -- $B > $O/${TAG}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_fetch -o run --output-format csv -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_write -o run --output-format csv -- $B > $O/${TAG}_write.log 2>&1
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds.csv
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq2 $O/${TAG}_sq_instruction_mix.csv
python3 scripts/summarize_profiles.py pmc $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_pmc_hbm_traffic.json
python3 scripts/summarize_profiles.py gbps $O/${TAG}_stats_serial.csv $O/${TAG}_pmc_hbm_traffic.json $O/${TAG}_hbm_gbps_per_kernel.csv
# the same SQ pass for the split-operand mode: MFMA-busy of its dominant kernels
rm -rf $O/${TAG}_sq
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_sq -o run --output-format csv -- $X2 > $O/${TAG}_sq_f16x2.log 2>&1
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds_f16x2.csv
for d in sq sq2 fetch write; do rm -rf $O/${TAG}_$d; done
fi
find $O/${TAG}_trace_overlap $O/${TAG}_trace_serial $O/${TAG}_trace_f16x2 $O/${TAG}_trace_c3 $O/${TAG}_trace_c2 $O/${TAG}_trace_rn -type f ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" -delete
find $O/${TAG}_trace_c3 $O/${TAG}_trace_c2 $O/${TAG}_trace_rn $O/${TAG}_trace_f16x2 -type f -name "*kernel_trace.csv" -delete
ls -la $O | grep ${TAG}
