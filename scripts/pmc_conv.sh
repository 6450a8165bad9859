#!/bin/bash
# LDS / issue counters of the convolution kernels at one shape, for each environment arm ("name=ENV..."), same box.
#   scripts/pmc_conv.sh "13 1024 1024 3" "cpt=Y2_HALO_COMPACT=1" "bord=Y2_HALO_COMPACT=0"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
shape="$1"; shift
mkdir -p gpurun_out/pmc
for arm in "$@"; do
  name="${arm%%=*}"; envs="${arm#*=}"
  rm -rf gpurun_out/pmc/$name
  for e in $envs; do export "$e"; done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU \
     -d gpurun_out/pmc/$name -o run --output-format csv -- python3 scripts/conv_one.py $shape 3 > gpurun_out/pmc/$name.log 2>&1
  for e in $envs; do unset "${e%%=*}"; done
  python3 scripts/summarize_profiles.py sq gpurun_out/pmc/$name gpurun_out/pmc/$name.csv > /dev/null 2>&1
  echo "== $name"; grep -E "kernel,|conv_halo|wgrad9" gpurun_out/pmc/$name.csv | cut -c1-400
  rm -rf gpurun_out/pmc/$name
done
