#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# usage: pmc_bf16_one.sh <outdir> M N K res tile   -- PMC passes over one bf16 1x1 conv shape (scripts/conv_one_bf16.py)
OUT=$GRAFT_REPO_ROOT/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/scripts/conv_one_bf16.py "$@" 2 > $OUT/pass$i.log 2>&1
  tail -1 $OUT/pass$i.log
done
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT conv_igemm_bf16
