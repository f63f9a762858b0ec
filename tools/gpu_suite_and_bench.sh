#!/bin/bash
# whole gpu suite (no -x), then the default bench line:  gpurun --timeout 2400 -- bash tools/gpu_suite_and_bench.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -40 | tee gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d['region_ms_per_step'])
for x in d['extra_configs']:
    print(x.get('key'), x.get('value'), x.get('error'))
PY
