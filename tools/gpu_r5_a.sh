#!/bin/bash
# round 5, session A: the tests touched by the bench / PPO / advisor items + PPO learning curves on the two BASELINE workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rf --no-header -p no:cacheprovider -k "gae or ppo or bench or precision" 2>&1 | grep -v amdgpu.ids | tail -25
for cfg in "myoHandPoseRandom-v0 4096 ${ITERS:-1000}" "myoFatiLegWalk-v0 1024 ${ITERS:-1000}"; do
  set -- $cfg
  timeout 600 python benchmarks/ppo_rollout.py --env $1 --num-envs $2 --iters $3 --skip-rollout-only --curve gpurun_out/ppo_curve_$1_$2.json $EXTRA 2>&1 | grep -v amdgpu.ids | tail -2
  python - <<PY
import json
d=json.load(open("gpurun_out/ppo_curve_$1_$2.json"))
w=d["windows"]
for x in w[:3]+w[len(w)//2-1:len(w)//2+1]+w[-3:]:
    print(x)
PY
done
