O=gpurun_out/run9; mkdir -p $O
python tests/tools/debug_c4_proposals.py > $O/debug_c4.txt 2>&1; cat $O/debug_c4.txt
bash scripts/micro/run8.sh > $O/run8_stdout.txt 2>&1; tail -60 $O/run8_stdout.txt
