#!/bin/bash
# Profile recipe of the ResNet-50 swap (run through gpurun from the repo root): kernel statistics and the HBM bytes per
# launch of its kernels (separate FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 correction), GB/s per kernel.
TAG=${1:-r03}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
W=2; K=3; ALL=$((W + K))
B="python3 bench.py --model resnet50 --batch 32 --steps $K --warmup $W --no-cpu-baseline"
O=gpurun_out
rm -rf $O/${TAG}_rn_trace $O/${TAG}_rn_fetch $O/${TAG}_rn_write
rocprofv3 --kernel-trace --stats -d $O/${TAG}_rn_trace -o run --output-format csv -- $B > $O/${TAG}_rn_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/${TAG}_rn_fetch -o run --output-format csv -- $B > $O/${TAG}_rn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/${TAG}_rn_write -o run --output-format csv -- $B > $O/${TAG}_rn_write.log 2>&1
python3 scripts/summarize_profiles.py stats $O/${TAG}_rn_trace $O/${TAG}_stats_resnet50_bs32.csv $ALL
python3 scripts/summarize_profiles.py pmc $O/${TAG}_rn_fetch $O/${TAG}_rn_write $O/${TAG}_pmc_hbm_traffic_resnet50.json
python3 scripts/summarize_profiles.py gbps $O/${TAG}_stats_resnet50_bs32.csv $O/${TAG}_pmc_hbm_traffic_resnet50.json $O/${TAG}_hbm_gbps_per_kernel_resnet50.csv
rm -rf $O/${TAG}_rn_fetch $O/${TAG}_rn_write
find $O/${TAG}_rn_trace -type f ! -name "*kernel_stats.csv" -delete
head -30 $O/${TAG}_hbm_gbps_per_kernel_resnet50.csv
