O=gpurun_out/run11; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "conv_bwd or train or bf16 or dp_gpu or config2 or config4" > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
python scripts/bench_train.py > $O/train_f32.json 2> $O/train_f32.err; cat $O/train_f32.json
python scripts/bench_train.py --bf16 > $O/train_bf16.json 2> $O/train_bf16.err; cat $O/train_bf16.json
