#!/bin/bash
# round-4 session H: rocprofv3 kernel statistics of the PPO iteration (fused learner kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_ppo
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ppo_hand -o hand -- python $R/benchmarks/ppo_rollout.py --env myoHandPoseRandom-v0 --num-envs 4096 --iters 10 > $R/gpurun_out/prof_ppo/hand.log 2>&1 < /dev/null
f=$(find /tmp/prof_ppo_hand -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $R/gpurun_out/prof_ppo/hand_kernel_stats.csv; head -16 "$f" | cut -c1-220; else echo "no kernel_stats.csv"; find /tmp/prof_ppo_hand | head; fi
grep "^{" $R/gpurun_out/prof_ppo/hand.log | cut -c1-400
