#!/bin/bash
# round-4 session D: the on-device PPO (ppo_rollout on the hand and the fati-leg, eager vs graphs) + its tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for a in "--env myoHandPoseRandom-v0 --num-envs 4096" "--env myoFatiLegWalk-v0 --num-envs 1024"; do
  timeout 600 python benchmarks/ppo_rollout.py $a --iters 10 2>&1 | grep -v amdgpu.ids | tail -3
  timeout 600 python benchmarks/ppo_rollout.py $a --iters 4 --eager 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/ppo_rollout.log
timeout 900 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "ppo or gae or struct_size or trailing" 2>&1 | grep -v amdgpu.ids | tail -15
