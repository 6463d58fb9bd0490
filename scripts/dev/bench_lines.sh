set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/round5b
mkdir -p $OUT
cd /tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $R/bench.py --f32-engine native --steps 50 --warmup 10 > $OUT/bench_native_f32_mfma.json 2> $OUT/bench_native.err
python3 $R/bench.py --f32-engine bf16x6 --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_bf16x6_exact_split.json 2> $OUT/bench_bf16x6.err
python3 $R/bench.py --steps 50 --warmup 10 --streams 1 > $OUT/bench_streams1.json 2> $OUT/bench_streams1.err
python3 $R/bench.py --steps 50 --warmup 10 --batch 1 --streams 12 --no-cpu-baseline > $OUT/bench_batch1_streams12.json 2> $OUT/bench_batch1_streams12.err
python3 $R/bench.py --config c1 --steps 50 --warmup 10 > $OUT/bench_c1_vgg16_rpn.json 2> $OUT/bench_c1.err
python3 $R/bench.py --config c4 --steps 50 --warmup 10 > $OUT/bench_c4_default.json 2> $OUT/bench_c4.err
python3 $R/bench.py --dtype bf16 --steps 50 --warmup 10 > $OUT/bench_c2_shapes_on_bf16.json 2> $OUT/bench_c2bf16.err
for f in bench_default bench_native_f32_mfma bench_bf16x6_exact_split bench_streams1 bench_batch1_streams12 bench_c1_vgg16_rpn bench_c4_default bench_c2_shapes_on_bf16; do python3 -c "
import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'])"; done
