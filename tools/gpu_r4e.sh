#!/bin/bash
# round-4 session E: the reference's PPO protocol (benchmarks/mjx_benchmark_PPO.py: 8192 envs, 5e6 steps, repeat 3) for its three env ids
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ppo
cd gpurun_out/ppo
for e in MjxElbowPoseRandom-v0 MjxFingerPoseRandom-v0 MjxHandReachRandom-v0; do
  timeout 900 python $R/benchmarks/mjx_benchmark_PPO.py --env_name $e --impl hip --num_envs 8192 2>&1 | grep -v amdgpu.ids | grep "Results for\|mean reward\|^{" 
done | tee $R/gpurun_out/mjx_benchmark_PPO.log
cd $R
python - <<'PY'
import json, re, glob, numpy as np
out = {"protocol": "benchmarks/mjx_benchmark_PPO.py of the reference: ppo_config of myosuite/envs/myo/mjx/__init__.py:43-67, num_timesteps 5e6, 8192 envs, timeit.repeat(number=1, repeat=3) around a full training run (graph capture included in each run)", "results": {}}
for f in sorted(glob.glob("gpurun_out/ppo/mjx_benchmark_PPO_results_*.npy")):
    r = np.load(f, allow_pickle=True).item()
    for k, v in r.items():
        out["results"][k] = {"seconds": [float(x) for x in v], "train_env_steps_per_s": [5e6 / float(x) for x in v]}
json.dump(out, open("gpurun_out/mjx_benchmark_PPO.json", "w"), indent=1)
print(json.dumps(out["results"], indent=1))
PY
timeout 900 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "ppo" 2>&1 | grep -v amdgpu.ids | tail -5
