GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for t in 71 76 74 72; do python3 scripts/conv_one.py 1 38 63 1024 2560 1 1 valid $t 30; done
