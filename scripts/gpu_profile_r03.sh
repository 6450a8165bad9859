#!/bin/bash
# Round-3 profile recipe (run through gpurun from the repo root): kernel trace + stats of the bench command with the
# production streams and serialised, SQ counter passes, FETCH / WRITE passes, HBM GB/s per kernel, and the LDS
# bank-conflict counters of conv_haloq with the bordered (default) and the compact (conflict-free) LDS image.
# rocprofv3 rule of this pool: the program goes directly after `--`; --pmc passes carry no other trace domain.
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=2; K=3; ALL=$((W + K))      # every traced step counts in the per-step columns
B="python3 bench.py --steps $K --warmup $W --no-cpu-baseline --no-f32-mode --kernel-events off --sustain-steps 0 --fed-steps 0"
O=gpurun_out
rm -rf $O/${TAG}_trace_overlap $O/${TAG}_trace_serial $O/${TAG}_sq $O/${TAG}_sq2 $O/${TAG}_fetch $O/${TAG}_write
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_overlap -o run --output-format csv -- $B > $O/${TAG}_trace_overlap.log 2>&1
export Y2_NO_WGRAD_OVERLAP=1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace_serial -o run --output-format csv -- $B > $O/${TAG}_trace_serial.log 2>&1
unset Y2_NO_WGRAD_OVERLAP
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/${TAG}_sq -o run --output-format csv -- $B > $O/${TAG}_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/${TAG}_sq2 -o run --output-format csv -- $B > $O/${TAG}_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_fetch -o run --output-format csv -- $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_write -o run --output-format csv -- $B > $O/${TAG}_write.log 2>&1
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_overlap $O/${TAG}_stats_overlap.csv $ALL
python3 scripts/summarize_profiles.py stats $O/${TAG}_trace_serial $O/${TAG}_stats_serial.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_overlap $O/${TAG}_timeline_overlap.csv $ALL
python3 scripts/summarize_profiles.py trace $O/${TAG}_trace_serial $O/${TAG}_timeline_serial.csv $ALL
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq $O/${TAG}_sq_busy_wait_lds.csv
python3 scripts/summarize_profiles.py sq $O/${TAG}_sq2 $O/${TAG}_sq_instruction_mix.csv
python3 scripts/summarize_profiles.py pmc $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_pmc_hbm_traffic.json
python3 scripts/summarize_profiles.py gbps $O/${TAG}_stats_serial.csv $O/${TAG}_pmc_hbm_traffic.json $O/${TAG}_hbm_gbps_per_kernel.csv
# LDS image A/B of conv_haloq at the two shapes VERDICT r2 names (13x13 512 -> 1024 and 1024 -> 1024), same box
( scripts/pmc_conv.sh "13 1024 1024 3" "compact=Y2_HALO_COMPACT=1" "bordered=Y2_HALO_COMPACT=0"; scripts/pmc_conv.sh "13 512 1024 3" "compact=Y2_HALO_COMPACT=1" "bordered=Y2_HALO_COMPACT=0" ) > $O/${TAG}_lds_image_compact_vs_bordered.txt 2>&1
for d in sq sq2 fetch write; do rm -rf $O/${TAG}_$d; done
find $O/${TAG}_trace_overlap $O/${TAG}_trace_serial -type f ! -name "*kernel_trace.csv" ! -name "*kernel_stats.csv" -delete
ls -la $O | grep ${TAG}
