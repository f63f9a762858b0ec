"""Simulation-throughput sweep with the protocol of the reference's benchmarks/mjx_benchmark.py:11-50: for
E in [64 .. 8192] envs, `loop_iterations` env-steps with fresh U[0,1) actions each step, 8192*loop_iterations total
steps per measurement, through the MJX-style functional API (myosuite_amd/mjx_api.py).

    python benchmarks/mjx_benchmark.py [--model hand|elbow]
"""
import argparse
import os
import sys
import timeit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from myosuite_amd import engine as E
from myosuite_amd.mjx_api import MjxPoseEnv


def measure_num_env_simulation_steps(model="hand", seed=0, loop_iterations=16):
    res = {}
    for e in [64, 512, 1024, 2048, 4096, 8192]:
        env = MjxPoseEnv(model=model, num_envs=e, seed=seed)
        state = env.reset(seed)
        act = torch.empty(e, env.action_size, device="cuda")

        def loop(state, key):
            for i in range(loop_iterations):
                E.uniform(act, seed=key, stream_id=i)
                state = env.step(state, act)
            return state

        state = loop(state, 0)          # preheat
        torch.cuda.synchronize()

        def run_benchmark():
            nonlocal state
            state = loop(state, 1)
            torch.cuda.synchronize()

        results = timeit.repeat(run_benchmark, number=8192 // e, repeat=3)
        print(f"Results for {e} envs: {8192 * loop_iterations} total steps take {results} seconds "
              f"({8192 * loop_iterations / np.mean(results):.3e} env-steps/s)")
        res[e] = float(np.mean(results))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hand")
    a = ap.parse_args()
    print(measure_num_env_simulation_steps(a.model))
