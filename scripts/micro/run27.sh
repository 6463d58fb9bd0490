#!/bin/bash
# lab evidence kept under profiles/round2_lab: the direct-to-LDS GEMM harness, weight-gradient tiles, bf16 head tiles
cd "$(dirname "$0")/../.."
O=gpurun_out/run27; mkdir -p $O
timeout 120 scripts/micro/_bin/bf16_lab > $O/bf16_lab.txt 2>&1
{ for b in 0 1; do echo "== FRCNN_WGRAD_BIG=$b, slice target 512 workgroups per layer (chip-filling grids)"; FRCNN_WGRAD_BIG=$b FRCNN_WGRAD_TARGET=1024 FRCNN_WGRAD_TARGET_BIG=512 python scripts/wgrad_time.py 2>/dev/null; done; } > $O/wgrad_time.txt
python scripts/conv_shapes.py --bf16 42,45,46,47,48 2>/dev/null | grep -v amdgpu > $O/conv_shapes_bf16_head_tiles.txt
python scripts/roi_bwd_time.py 2>/dev/null > $O/roi_bwd_time.txt
for f in $O/*.txt; do tail -n 3 $f; done
