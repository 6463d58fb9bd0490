#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/run23; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
python scripts/bench_train.py --steps 40 --warmup 8 > $O/bench_train_f32.json 2>/dev/null; tail -1 $O/bench_train_f32.json | cut -c1-1200
python scripts/bench_train.py --bf16 --steps 40 --warmup 8 > $O/bench_train_mixed_bf16.json 2>/dev/null; tail -1 $O/bench_train_mixed_bf16.json | cut -c1-1200
