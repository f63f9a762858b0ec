"""What the host-buffer boundaries cost ("PCIe-inclusive").

`bench.py`'s `value` is measured with actions and state resident in HBM (the C ABI takes device pointers).  The gym / vector
boundary of the reference (`env.step(np.ndarray) -> np.ndarray`, myosuite/envs/env_base.py:282-356, SB3's VecEnv) hands HOST
buffers over, so a CPU-side learner pays a PCIe round trip per env-step.  This script times, on one workload and one GPU:

  device        rollout_step(None): in-kernel action draw, nothing leaves HBM               (= bench.py's region)
  device_action rollout_step(device action tensor)
  host_pinned   MyoVecEnv.step_host(np actions): pinned H2D + launch + 3 pinned D2H + 1 sync (the PCIe-inclusive rate)
  host_step5    MyoVecEnv.step5: gym 5-tuple through the dict-carrying step(), pageable .cpu() copies
  host_sb3      MyoVecEnv.step: SB3 protocol (list of per-env info dicts built in Python)

    python benchmarks/host_boundary.py [--env myoHandPoseRandom-v0] [--num-envs 4096] [--steps 200] [--out gpurun_out/host_boundary.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from myosuite_amd import gym_compat as mg          # noqa: E402


def _time(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="myoHandPoseRandom-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    n = a.num_envs
    vec = mg.MyoVecEnv(a.env, n, seed=0)
    env = vec.env
    env.rollout_setup(action_seed=0)
    nu, od = env.cm.nu, env.obs.shape[1]
    act_dev = torch.rand(n, nu, device=env.device)
    act_np = np.random.default_rng(0).random((n, nu), dtype=np.float32)
    k = [0]

    def dev():
        env.rollout_step(None, stream_id=k[0]); k[0] += 1
    rows = {}
    rows["device"] = _time(dev, a.steps, a.warmup)
    rows["device_action"] = _time(lambda: env.rollout_step(act_dev), a.steps, a.warmup)
    rows["host_pinned"] = _time(lambda: vec.step_host(act_np), a.steps, a.warmup)
    rows["host_step5"] = _time(lambda: vec.step5(act_np), max(10, a.steps // 4), 3)
    rows["host_sb3"] = _time(lambda: vec.step(act_np), max(10, a.steps // 4), 3)
    h2d, d2h = n * nu * 4, n * (od + env.rwd.shape[1]) * 4 + n
    out = {"what": "seconds per env-step of one batch through each boundary; env-steps/s = num_envs / that", "env": a.env, "num_envs": n,
           "steps": a.steps, "bytes_per_step": {"h2d_actions": h2d, "d2h_obs_reward_rows_done": d2h},
           "device": torch.cuda.get_device_name(0),
           "paths": {kk: {"ms_per_step": 1e3 * v, "env_steps_per_s": n / v} for kk, v in rows.items()}}
    out["pcie_inclusive_over_device_resident"] = rows["device"] / rows["host_pinned"]
    out["pcie_bytes_per_s_on_host_pinned"] = (h2d + d2h) / rows["host_pinned"]
    print(json.dumps(out, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
