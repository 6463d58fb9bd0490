#!/bin/bash
# round 6: the bf16 engine on v_mfma_f32_16x16x32_bf16 -- its tests, then configs[3] and the mixed training steps
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python -m pytest tests/test_bf16_gpu.py tests/test_conv_bwd_gpu.py tests/test_train_graph_gpu.py -x -q 2>&1 | grep -v "^frame\|amdgpu.ids" | tail -12
python -m pytest tests/test_configs_full_size_gpu.py -x -q -k "config3 or config4 or bf16 or mixed" 2>&1 | tail -4
python bench.py --config c4 --no-cpu-baseline --no-io > gpurun_out/r6_c4.json 2> gpurun_out/r6_c4.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6_c4.json"))
print("c4", d["value"], "frac", d["roofline"]["frac"], d["roofline"].get("kernel"))
PY
python scripts/bench_train.py --bf16 --steps 60 --warmup 40 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('mixed steps', d['rpn_step1']['ms_per_step'], d['det_step2']['ms_per_step'])"
