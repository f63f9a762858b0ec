#!/bin/bash
# round-4 session F: forward carry -- bit identity tests, then the bench lines it touches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "forward_carry or folded or two_wave or implicit" 2>&1 | grep -v amdgpu.ids | tail -15
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d['region_ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])
for x in d['extra_configs']:
    print(x.get('key'), x.get('value'), x.get('error'))
PY
