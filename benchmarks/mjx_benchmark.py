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


def measure_num_env_simulation_steps(model="hand", seed=0, loop_iterations=16, graph=True):
    """graph=True: the 16-step loop is captured once into a HIP graph and replayed -- the counterpart of the reference's
    jitted ``jax.lax.scan`` over 16 steps (mjx_benchmark.py:24-33); actions are drawn inside the graph (torch's graph-safe
    Philox), as the reference draws them inside the scan.  graph=False launches every step eagerly."""
    res = {}
    for e in [64, 512, 1024, 2048, 4096, 8192]:
        env = MjxPoseEnv(model=model, num_envs=e, seed=seed)
        state = env.reset(seed)
        act = torch.empty(e, env.action_size, device="cuda")

        def loop(state, key):
            for i in range(loop_iterations):
                if graph:
                    act.uniform_()
                else:
                    E.uniform(act, seed=key, stream_id=i)
                state = env.step(state, act)
            return state

        state = loop(state, 0)          # preheat
        torch.cuda.synchronize()
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                state = loop(state, 0)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                state = loop(state, 1)

        def run_benchmark():
            nonlocal state
            if graph:
                g.replay()
            else:
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
    ap.add_argument("--eager", action="store_true", help="launch every step instead of replaying a captured 16-step HIP graph")
    a = ap.parse_args()
    print(measure_num_env_simulation_steps(a.model, graph=not a.eager))
