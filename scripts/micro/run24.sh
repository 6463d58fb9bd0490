#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/run24; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
run() { echo "== $*"; python scripts/bench_train.py --steps 40 --warmup 8 "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print({k:(v['ms_per_step'] if isinstance(v,dict) and 'ms_per_step' in v else None) for k,v in d.items() if isinstance(v,dict)})"; }
run
run --bf16
run --sync-each-step
run --bf16 --sync-each-step
