O=gpurun_out/run7; mkdir -p $O
python -m pytest tests -m gpu -q -s -k "config3 or weights_are_current or adam_one_step or mixed_precision_training_step_against" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -8 $O/pytest.txt
python scripts/bench_train.py > $O/train_f32.json 2> $O/train_f32.err; cat $O/train_f32.json
python scripts/bench_train.py --bf16 > $O/train_bf16.json 2> $O/train_bf16.err; cat $O/train_bf16.json
python bench.py --steps 50 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err; tail -c 2500 $O/bench_default.json
python bench.py --steps 50 --warmup 10 --streams 1 --no-cpu-baseline > $O/bench_streams1.json 2> $O/bench_streams1.err; tail -c 1200 $O/bench_streams1.json
python bench.py --config c4 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; tail -c 1500 $O/bench_c4.json
python scripts/bench_vgg.py > $O/bench_vgg.txt 2>&1; tail -5 $O/bench_vgg.txt
