O=gpurun_out/suite; mkdir -p $O
python -m pytest tests -m gpu -q --durations=8 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -14 $O/pytest.txt
