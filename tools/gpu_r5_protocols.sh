#!/bin/bash
# the reference's two benchmark protocols at the round-5 HEAD: benchmarks/mjx_benchmark.py (three ids, six batch sizes) and
# benchmarks/mjx_benchmark_PPO.py (three ids, 8192 envs, 5e6 steps, timeit.repeat(repeat=3))
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/proto; cd gpurun_out/proto
timeout 900 python $R/benchmarks/mjx_benchmark.py --json $R/gpurun_out/r05_mjx_benchmark.json 2>&1 < /dev/null | grep -v amdgpu.ids | tail -25 | tee $R/gpurun_out/r05_mjx_benchmark.log
for e in MjxElbowPoseRandom-v0 MjxFingerPoseRandom-v0 MjxHandReachRandom-v0; do
  timeout 600 python $R/benchmarks/mjx_benchmark_PPO.py --env_name $e --impl hip --num_envs 8192 2>&1 < /dev/null | grep -v amdgpu.ids | grep "Results for\|mean reward"
done | tee $R/gpurun_out/r05_mjx_benchmark_PPO.log
ls
