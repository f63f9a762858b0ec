#!/bin/bash
# Round-3 evidence session: gpu tests, rocprofv3 profiles of the seven bench workloads, default bench line, host CPU scaling curve.
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r03b}
cd $R; mkdir -p gpurun_out; rm -f gpurun_out/full_batch_stage_errors.json
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" gpurun_out/pytest_gpu.log | tail -4
bash tools/gpu_prof_all.sh $TAG | tail -3
timeout 600 python tools/cpu_scaling.py --seconds 3 --out gpurun_out/cpu_scaling_hand.json > /dev/null 2> gpurun_out/cpu_scaling.err; tail -10 gpurun_out/cpu_scaling.err
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'])
for e in d['extra_configs']: print(e.get('key'), round(e.get('value',0)), e.get('roofline',{}).get('kernel_ms'), e.get('roofline',{}).get('traffic'), e.get('error'))
print(d['cpu_baseline'])"
