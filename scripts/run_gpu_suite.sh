#!/bin/bash
# GPU box: the -m gpu suite, then the headline bench with 8 images in flight and with one
cd "$(dirname "$0")/.."
O=gpurun_out/suite; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
python bench.py --steps 50 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
python bench.py --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err; tail -c 1500 $O/bench_streams1.json
