#!/bin/bash
# Host-side AddressSanitizer + UBSan pass over the C-ABI library's host code.  CPU box; no GPU needed.
#   bash scripts/sanitize_host.sh [out.txt]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-profiles/round4_host_sanitizers.txt}
B=/tmp/frcnn_sanitize; mkdir -p $B
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
pids=""
for f in faster_rcnn_amd/csrc/*.hip; do
  o=$B/$(basename $f .hip).o
  EXTRA=""; case $f in *roi.hip|*boxes.hip|*detect.hip|*sort_nms.hip) EXTRA="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f -o $o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -c tests/tools/host_sanitize.cpp -o $B/driver.obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined -fno-gpu-sanitize $B/*.o $B/driver.obj -o $B/host_sanitize
{ echo "# hipcc -fsanitize=address,undefined -fno-gpu-sanitize (host code of every csrc/*.hip) + tests/tools/host_sanitize.cpp"; 
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $B/host_sanitize 2>&1 | grep -v "amdgpu.ids" ; echo "exit code ${PIPESTATUS[0]}"; } | tee $OUT
