for T in 2048 1024 512 256; do TAG="target=$T" FRCNN_WGRAD_TARGET=$T python scripts/micro/train_ab2.py 2>/dev/null | tail -1; done
