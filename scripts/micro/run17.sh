#!/bin/bash
# bf16 main-loop A/B (FRCNN_BF16_VARIANT 0 = round-1 loop, 2 = mid-chunk barrier) + the gpu suite on the current tree
cd "$(dirname "$0")/../.."
O=gpurun_out/run17; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
for v in 0 2 0 2; do
  echo "== bf16 variant $v"
  FRCNN_BF16_VARIANT=$v python bench.py --config c4 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('c4', d['value'], d['ms_per_step'])"
  FRCNN_BF16_VARIANT=$v python scripts/bench_train.py --bf16 --steps 30 --warmup 8 2>/dev/null | tail -2 | cut -c1-400
done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-900
