GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $GRAFT_REPO_ROOT
for t in 0 77 74; do
  python3 scripts/conv_one.py 1 149 249 64 64 3 1 same $t 50
  python3 scripts/conv_one.py 1 600 1000 64 64 3 1 same $t 20
  python3 scripts/conv_one.py 1 300 500 128 128 3 1 same $t 20
  python3 scripts/conv_one.py 1 149 249 256 64 1 1 valid $t 50
done
