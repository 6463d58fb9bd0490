#!/bin/bash
# (The two macros this script builds with -- H3_LAB_A_ONCE, H3_PF2 -- were taken out of csrc/conv_h3.hip again: commit 71aa159 holds them.)
# Lab: what would the head's 3x3 layer cost if a channel chunk of activations were staged in LDS ONCE and its nine taps read it there
# (row-shifted fragment reads) instead of nine fetches from L2?  Builds a second library with -DH3_LAB_A_ONCE (activations fetched for
# the first tap of a chunk only: wrong products, right amount of every other work) -- here, on the CPU box -- and times both on the GPU.
#   build (CPU box):  bash scripts/dev/r6_a_once_lab.sh build      run (GPU box): bash scripts/dev/r6_a_once_lab.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$R/scripts/micro/_bin
if [ "${1:-run}" = build ]; then
  O=$R/faster_rcnn_amd/csrc/_obj
  objs=$(ls $O/*.o | grep -v conv_h3.o)
  for v in a_once:-DH3_LAB_A_ONCE pf2:-DH3_PF2; do
    /opt/rocm/bin/hipcc -c $R/faster_rcnn_amd/csrc/conv_h3.hip -o /tmp/conv_h3_${v%%:*}.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 ${v##*:} || exit 1
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-z,defs -o $B/libfrcnn_hip_${v%%:*}.so $objs /tmp/conv_h3_${v%%:*}.o && ls -la $B/libfrcnn_hip_${v%%:*}.so
  done
  exit
fi
cd $R
for rep in 1 2; do
  for lib in product a_once pf2; do
    if [ $lib != product ]; then export FRCNN_LIB_PATH=$B/libfrcnn_hip_$lib.so; else unset FRCNN_LIB_PATH; fi
    python3 scripts/dev/r6_head_3x3.py $lib
  done
done
