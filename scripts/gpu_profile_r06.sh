#!/bin/bash
# Round-6 profile recipe (run through gpurun from the repo root, on the round's LAST binary: VERDICT r5 next 7).
# As gpu_profile_r05.sh for the headline (C4, f16) command -- kernel trace + stats with the production streams and
# serialised, SQ / FETCH / WRITE counter passes, per-layer tables -- plus
#   * the reference-tolerance modes on the f16 pipe: f16x2f (round 6: split forward, single-product backward) and f16x2:
#     trace + stats + timeline + per-layer table, one SQ pass each, and -- new -- their own FETCH_SIZE / WRITE_SIZE passes
#     (wasted-traffic ratio of the split modes);
#   * the ResNet swap's kernel table of one applied, graph-replayed step (batch 32), C2 / C3 kernel stats.
# rocprofv3 rule of this pool: the program goes directly after `--`; --pmc passes carry no other trace domain.
TAG=${1:-r06}
PMC=${2:-1}          # 0: skip the counter passes
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=2; K=3; ALL=$((W + K))
COMMON="--steps $K --warmup $W --no-cpu-baseline --no-f32-mode --no-fast-parity-mode --no-extra-legs --kernel-events off --sustain-steps 0 --fed-steps 0"
B="python3 bench.py $COMMON"
XF="python3 bench.py --dtype f16x2f $COMMON"
X2="python3 bench.py --dtype f16x2 $COMMON"
C3="python3 bench.py --model classifier"
C2="python3 bench.py --forward-only --batch 32 --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --kernel-events off --sustain-steps 0"
RN="python3 bench.py --model resnet50 --batch 32 --graph --steps 6 --warmup 4 --no-cpu-baseline"
O=gpurun_out
for d in trace_overlap trace_serial trace_f16x2f trace_f16x2 trace_c3 trace_c2 trace_rn sq sq2 fetch write; do rm -rf $O/${TAG}_$d; done
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_overlap -o run --output-format csv -- $B > $O/${TAG}_trace_overlap.log 2>&1
export Y2_NO_WGRAD_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_serial -o run --output-format csv -- $B > $O/${TAG}_trace_serial.log 2>&1
unset Y2_NO_WGRAD_OVERLAP
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_f16x2f -o run --output-format csv -- $XF > $O/${TAG}_trace_f16x2f.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_f16x2 -o run --output-format csv -- $X2 > $O/${TAG}_trace_f16x2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_c3 -o run --output-format csv -- $C3 > $O/${TAG}_trace_c3.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_c2 -o run --output-format csv -- $C2 > $O/${TAG}_trace_c2.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_rn -o run --output-format csv -- $RN > $O/${TAG}_trace_rn.log 2>&1
S="python3 scripts/summarize_profiles.py"
$S stats $O/${TAG}_trace_overlap $O/${TAG}_stats_overlap.csv $ALL
$S stats $O/${TAG}_trace_serial $O/${TAG}_stats_serial.csv $ALL
$S trace $O/${TAG}_trace_overlap $O/${TAG}_timeline_overlap.csv $ALL
$S trace $O/${TAG}_trace_serial $O/${TAG}_timeline_serial.csv $ALL
for m in f16x2f f16x2; do
  $S stats $O/${TAG}_trace_$m $O/${TAG}_stats_$m.csv $ALL
  $S trace $O/${TAG}_trace_$m $O/${TAG}_timeline_$m.csv $ALL
done
$S stats $O/${TAG}_trace_c3 $O/${TAG}_stats_c3.csv 23        # 3 warm-up + 20 timed steps
$S stats $O/${TAG}_trace_c2 $O/${TAG}_stats_c2_forward.csv $ALL
# one APPLIED graph-replayed step (bench settles the loss scale first): the dispatches between the last two root convolutions
$S step $O/${TAG}_trace_rn $O/${TAG}_resnet50_step.csv rn_conv7_mfma_kernel
python3 scripts/profile_layers.py > $O/${TAG}_layers_per_layer_us.txt 2>&1
DTYPE=f16x2f python3 scripts/profile_layers.py > $O/${TAG}_layers_f16x2f.txt 2>&1
DTYPE=f16x2 python3 scripts/profile_layers.py > $O/${TAG}_layers_f16x2.txt 2>&1
DTYPE=f32 python3 scripts/profile_layers.py > $O/${TAG}_layers_f32.txt 2>&1
MODEL=classifier python3 scripts/profile_layers.py > $O/${TAG}_layers_c3.txt 2>&1
if [ "$PMC" = "1" ]; then
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $SQ1 -d $O/${TAG}_sq -o run --output-format csv -- $B > $O/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/${TAG}_sq2 -o run --output-format csv -- $B > $O/${TAG}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_fetch -o run --output-format csv -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_write -o run --output-format csv -- $B > $O/${TAG}_write.log 2>&1
$S sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds.csv
$S sq $O/${TAG}_sq2 $O/${TAG}_sq_instruction_mix.csv
$S pmc $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_pmc_hbm_traffic.json
$S gbps $O/${TAG}_stats_serial.csv $O/${TAG}_pmc_hbm_traffic.json $O/${TAG}_hbm_gbps_per_kernel.csv
# the reference-tolerance modes: one SQ pass and their own FETCH / WRITE passes (separate runs, as the guide prescribes)
for m in f16x2f f16x2; do
  CMD=$XF; [ "$m" = "f16x2" ] && CMD=$X2
  for d in sq fetch write; do rm -rf $O/${TAG}_$d; done
  rocprofv3 --pmc $SQ1 -d $O/${TAG}_sq -o run --output-format csv -- $CMD > $O/${TAG}_sq_$m.log 2>&1
  rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_fetch -o run --output-format csv -- $CMD > $O/${TAG}_fetch_$m.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_write -o run --output-format csv -- $CMD > $O/${TAG}_write_$m.log 2>&1
  $S sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds_$m.csv
  # (file name without "_pmc_hbm_traffic": bench.py's roofline.traffic reads the HEADLINE command's pass only)
  $S pmc $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_hbm_traffic_$m.json
  $S gbps $O/${TAG}_stats_$m.csv $O/${TAG}_hbm_traffic_$m.json $O/${TAG}_hbm_gbps_per_kernel_$m.csv
done
for d in sq sq2 fetch write; do rm -rf $O/${TAG}_$d; done
fi
find $O/${TAG}_trace_* -type f ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" -delete
find $O/${TAG}_trace_c3 $O/${TAG}_trace_c2 $O/${TAG}_trace_rn $O/${TAG}_trace_f16x2 $O/${TAG}_trace_f16x2f -type f -name "*kernel_trace.csv" -delete
ls -la $O | grep ${TAG}
