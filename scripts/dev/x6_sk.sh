GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for t in 0 74 274 374 474 674; do
  python3 scripts/conv_one.py 1 38 63 1024 256 1 1 valid $t 50
  python3 scripts/conv_one.py 1 38 63 512 256 1 1 valid $t 50
done
