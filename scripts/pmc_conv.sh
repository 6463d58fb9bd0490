#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# usage: pmc_conv.sh <outdir> <conv_one args...>; separate --pmc passes (guide: no mixing with traces)
OUT=$1; shift
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pass$i -- python3 $GRAFT_REPO_ROOT/scripts/conv_one.py "$@" > $GRAFT_REPO_ROOT/$OUT.pass$i.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/$OUT.pass$i.log
done
