O=gpurun_out/run12; mkdir -p $O
for F in 1000 12 8 5 3; do
  echo "== FRCNN_WGRAD_FLUSH=$F" >> $O/train.txt
  FRCNN_WGRAD_FLUSH=$F python scripts/bench_train.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 ', d['rpn_step1_ms'], d['det_step2_ms'])" >> $O/train.txt
  FRCNN_WGRAD_FLUSH=$F python scripts/bench_train.py --bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['rpn_step1_ms'], d['det_step2_ms'])" >> $O/train.txt
done
cat $O/train.txt
