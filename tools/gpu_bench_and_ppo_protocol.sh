#!/bin/bash
# the bench bookkeeping test, the default bench line (incl. ppo_training), the reference's PPO protocol at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ppo
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q -rf --no-header -p no:cacheprovider 2>&1 < /dev/null | grep -v amdgpu.ids | tail -8
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err < /dev/null; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().split('\n')[-1])
print('headline', d['value'], d['ms_per_step'], d['region_ms_per_step'])
for x in d['extra_configs']:
    print(x.get('key'), x.get('value'), x.get('error'))
for x in d['ppo_training']:
    print(x.get('key'), x.get('train_env_steps_per_s'), x.get('rollout_env_steps_per_s'), x.get('error'))
PY
cd gpurun_out/ppo
for e in MjxElbowPoseRandom-v0 MjxFingerPoseRandom-v0 MjxHandReachRandom-v0; do
  timeout 600 python $R/benchmarks/mjx_benchmark_PPO.py --env_name $e --impl hip --num_envs 8192 2>&1 < /dev/null | grep -v amdgpu.ids | grep "Results for\|mean reward"
done | tee $R/gpurun_out/mjx_benchmark_PPO.log
