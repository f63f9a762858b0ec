#!/bin/bash
# round-4 session J: fused PPO kernels -- tests, ppo_rollout, kernel statistics
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/ppo gpurun_out/prof_ppo
timeout 300 python -m pytest tests/test_ppo_fused.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 < /dev/null | grep -v amdgpu.ids | tail -5
for a in "--env myoHandPoseRandom-v0 --num-envs 4096" "--env myoFatiLegWalk-v0 --num-envs 1024"; do
  timeout 300 python benchmarks/ppo_rollout.py $a --iters 20 2>&1 < /dev/null | grep -v amdgpu.ids | tail -1 | cut -c1-420
done | tee gpurun_out/ppo_rollout.log
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ppo_hand -o hand -- python $R/benchmarks/ppo_rollout.py --env myoHandPoseRandom-v0 --num-envs 4096 --iters 10 > $R/gpurun_out/prof_ppo/hand.log 2>&1 < /dev/null)
f=$(find /tmp/prof_ppo_hand -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/prof_ppo/hand_kernel_stats.csv; head -8 "$f" | cut -c1-140; fi
