#!/bin/bash
# round 6: the f16x3 fences -- new tests, the engine's existing tests, then the headline (cost of the running maximum in the loaders)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
python -m pytest tests/test_h3_fences_gpu.py -x -q -s 2>&1 | grep -v "^frame\|amdgpu.ids" | tail -40
python -m pytest tests/test_conv_h3_gpu.py tests/test_abi_host_gpu.py -x -q 2>&1 | tail -5
python bench.py --no-extra --no-cpu-baseline --no-io > gpurun_out/r6_fence_bench.json 2> gpurun_out/r6_fence_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r6_fence_bench.json"))
print("headline", d["value"], "frac", d["roofline"]["frac"], "native", d.get("native_f32_mfma",{}).get("value"))
PY
