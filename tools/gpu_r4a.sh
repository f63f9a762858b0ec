#!/bin/bash
# round-4 session A: gpu tests, the reference's simulation sweep (benchmarks/mjx_benchmark.py), default bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -x 2>&1 | grep -v amdgpu.ids | tail -25 | tee gpurun_out/pytest_gpu.log
timeout 900 python benchmarks/mjx_benchmark.py --json gpurun_out/mjx_benchmark.json --out gpurun_out/mjx_benchmark_results.npy 2>&1 | grep -v amdgpu.ids | grep "Results\|MyoSuite" | tee gpurun_out/mjx_benchmark.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d['region_ms_per_step'], d['gpu_clocks_mhz'])
print('profile', d['roofline']['profile'])
for x in d['extra_configs']:
    print(x.get('key'), x.get('value'), x.get('error'), (x.get('roofline') or {}).get('launch'))
PY
