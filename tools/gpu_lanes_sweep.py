"""Throughput of one env id at every compiled group width:  python tools/gpu_lanes_sweep.py ENV_ID NENV"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.envs import registry
env_id, n = sys.argv[1], int(sys.argv[2])
for lanes in (4, 8, 16, 32, 64):
    try:
        env = registry.make(env_id, num_envs=n, seed=0, lanes_per_env=lanes)
    except Exception as exc:
        print(lanes, "n/a", str(exc)[:60]); continue
    env.rollout_setup(action_seed=0)
    for s in range(6): env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    K = 48
    t0 = time.perf_counter()
    for s in range(K): env.rollout_step(None, stream_id=6 + s)
    torch.cuda.synchronize()
    print(f"{env_id} n={n} G={lanes:2d} {n * K / (time.perf_counter() - t0) / 1e6:8.3f} M env-steps/s")
