O=gpurun_out/run14; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "train or bf16 or dp_gpu or config2 or config4 or bench_dp" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
for G in 1 0; do
  echo "== FRCNN_TRAIN_GRAPH=$G" >> $O/train.txt
  FRCNN_TRAIN_GRAPH=$G python scripts/bench_train.py 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 ', d['rpn_step1_ms'], d['det_step2_ms'])" >> $O/train.txt
  FRCNN_TRAIN_GRAPH=$G python scripts/bench_train.py --bf16 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', d['rpn_step1_ms'], d['det_step2_ms'])" >> $O/train.txt
done
cat $O/train.txt; tail -5 $O/err.txt
