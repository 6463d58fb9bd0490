O=gpurun_out/run10; mkdir -p $O
scripts/micro/_bin/conv_lab time train 0,123,223,323,423,623 > $O/time_train.txt 2>&1; cat $O/time_train.txt
python -m pytest tests -m gpu -q -s -k "config3" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
