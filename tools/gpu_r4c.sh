#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tests/tools/gpu_r4_debug.py 2>&1 | grep -v amdgpu.ids | grep -v "e-0[6789] status 0$" | tee gpurun_out/r4_debug.log
timeout 1800 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | grep -v "^E   \|^    \|^$" | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d['region_ms_per_step'])
for x in d['extra_configs']:
    print(x.get('key'), x.get('value'), x.get('error'))
PY
