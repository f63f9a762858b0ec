#!/bin/bash
# round-4 session G: the fused PPO learner kernels -- ppo_rollout (hand@4096, fati-leg@1024: fused / torch learner / brax nets), then the
# reference's PPO protocol for its three env ids
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ppo
for a in "--env myoHandPoseRandom-v0 --num-envs 4096" "--env myoFatiLegWalk-v0 --num-envs 1024"; do
  timeout 600 python benchmarks/ppo_rollout.py $a --iters 20 2>&1 | grep -v amdgpu.ids | tail -1
  MYOSIM_PPO_SAMPLES=16 timeout 600 python benchmarks/ppo_rollout.py $a --iters 20 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python benchmarks/ppo_rollout.py $a --iters 10 --torch-learner 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 600 python benchmarks/ppo_rollout.py $a --iters 10 --nets brax 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/ppo_rollout.log
cd gpurun_out/ppo
for e in MjxElbowPoseRandom-v0 MjxFingerPoseRandom-v0 MjxHandReachRandom-v0; do
  timeout 900 python $R/benchmarks/mjx_benchmark_PPO.py --env_name $e --impl hip --num_envs 8192 2>&1 | grep -v amdgpu.ids | grep "Results for\|mean reward\|^{"
done | tee $R/gpurun_out/mjx_benchmark_PPO.log
