#!/bin/bash
# Dev tool (through gpurun): SQ counters of the 1x1 kernels on the three configs[3] shapes (scripts/diag_1x1.py), conv_igemm
# (Y2_GEMM1=0) against conv_gemm1 (Y2_GEMM1=1).  Each --pmc pass is its own run (pool rule), each under its own `timeout`.
# (TA_* / TCC_* passes aborted inside rocprofv3 on this pool and hung the call for 25 minutes: SQ counters only.)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out
for g in ${GEMM1_MODES:-0 1}; do
  export Y2_GEMM1=$g
  for p in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
    tag=$(echo $p | cut -d' ' -f1)
    rm -rf $O/d1x1_pmc
    timeout 120 rocprofv3 --pmc $p -d $O/d1x1_pmc -o run --output-format csv -- python3 scripts/diag_1x1.py > $O/d1x1_g${g}_$tag.log 2>&1
    python3 scripts/summarize_profiles.py sq $O/d1x1_pmc $O/d1x1_g${g}_$tag.csv > /dev/null 2>&1
  done
done
rm -rf $O/d1x1_pmc
