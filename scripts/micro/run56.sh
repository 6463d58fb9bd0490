O=gpurun_out/suite; mkdir -p $O
python -m pytest tests -m gpu -q -s --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -40 $O/pytest.txt
