#!/bin/bash
cd "$(dirname "$0")/../.."
python -m pytest tests/test_conv_bwd_gpu.py -m gpu -q -x 2>&1 | tail -2
for t in 64 256 512; do echo "== FRCNN_WGRAD_BIG=1 target $t"; FRCNN_WGRAD_TARGET_BIG=$t python scripts/wgrad_time.py 2>/dev/null; done
echo "== 64x64 kernel target 512";  FRCNN_WGRAD_BIG=0 FRCNN_WGRAD_TARGET=1024 python scripts/wgrad_time.py 2>/dev/null
