#!/bin/bash
# PMC passes (separate runs, --pmc only) over the fp32 training steps: matrix-pipe busy and HBM-side bytes of the weight-gradient and
# split-K kernels of the split engine.   usage: bash scripts/pmc_train.sh [out dir under gpurun_out]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-pmc_train}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $ctrs --output-format csv -d $OUT/p$i -- python3 $R/scripts/bench_train.py --steps 4 --warmup 2 --sync-each-step > $OUT/p$i.log 2>&1
done
{
for pat in "k_conv_wgrad_batch<4>" "k_conv_igemm_x6<2, 1, 2, 4, true>" "k_conv_igemm_x6<1, 1, 2, 2, true>" "k_conv_igemm_x6<1, 1, 2, 2, false>" "k_pack_x6_batch" "k_wgrad_reduce_batch"; do
  echo "== $pat (means per dispatch over the fp32 RPN + detector steps)"
  for p in 1 2 3 4; do python3 $R/scripts/pmc_summary.py $OUT/p$p "$pat"; done
done
} > $OUT/summary.txt 2>&1
find $OUT -name "*counter_collection.csv" -size +10M -delete
cat $OUT/summary.txt
