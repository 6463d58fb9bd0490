GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
export FRCNN_BENCH_NO_ENTRY=1 FRCNN_BENCH_NO_NATIVE=1
run() { python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-io "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*:', d['value'], 'img/s')"; }
run
run --split-k off
C="res5a_branch2c=76,res5b_branch2c=76,res5c_branch2c=76"
run --unit-tiles $C
B="res5a_branch2b=71,res5b_branch2b=71,res5c_branch2b=71,res5b_branch2a=71,res5c_branch2a=71"
run --unit-tiles $B
C2="res5a_branch2c=72,res5b_branch2c=72,res5c_branch2c=72"
run --unit-tiles $C2
run
