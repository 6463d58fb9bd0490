#!/bin/bash
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
# GPU box: time + HBM-side traffic of the head's 128x128 launches under FRCNN_GROUP_M (tile order inside an XCD's run).
# usage: bash scripts/group_m_sweep.sh <outdir>
OUT=$GRAFT_REPO_ROOT/$1; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for shape in "300 7 7 512 512 3 1 same 121 30 1" "300 7 7 2048 512 1 1 valid 126 30 1"; do
  tag=$(echo $shape | tr ' ' '_')
  for g in 0 1 2 4 8; do
    export FRCNN_GROUP_M=$g
    python3 $GRAFT_REPO_ROOT/scripts/conv_one.py $shape > $OUT/time_${tag}_g$g.txt 2>&1
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --output-format csv -d $OUT/pmc_${tag}_g${g}_$ctr -- python3 $GRAFT_REPO_ROOT/scripts/conv_one.py $shape > /dev/null 2>&1
    done
    f=$(python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT/pmc_${tag}_g${g}_FETCH_SIZE conv_igemm | grep FETCH_SIZE)
    w=$(python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $OUT/pmc_${tag}_g${g}_WRITE_SIZE conv_igemm | grep WRITE_SIZE)
    echo "shape=[$shape] group_m=$g :: $(cat $OUT/time_${tag}_g$g.txt | tail -1) :: $f :: $w" | tee -a $OUT/summary.txt
    rm -rf $OUT/pmc_${tag}_g${g}_FETCH_SIZE $OUT/pmc_${tag}_g${g}_WRITE_SIZE
  done
done
