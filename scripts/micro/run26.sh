#!/bin/bash
cd "$(dirname "$0")/../.."
run() { echo "== $*"; env "$@" python scripts/bench_train.py --bf16 --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print({k:(v['ms_per_step'] if isinstance(v,dict) and 'ms_per_step' in v else None) for k,v in d.items() if isinstance(v,dict)})"; }
run FRCNN_BF16_BIG=42
run FRCNN_BF16_BIG=47
run FRCNN_BF16_BIG=42
run FRCNN_BF16_BIG=47
