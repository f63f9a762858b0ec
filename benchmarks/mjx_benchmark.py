"""Simulation-throughput sweep with the protocol of the reference's benchmarks/mjx_benchmark.py:11-62: for each of the three
env ids the reference sweeps (MjxElbowPoseRandom-v0, MjxFingerPoseRandom-v0, MjxHandReachRandom-v0) and E in [64 .. 8192] envs,
`loop_iterations` = 16 env-steps with fresh U[0,1) actions each step, 8192 * 16 total steps per measurement
(timeit.repeat(number = 8192 // E, repeat = 3), mean reported), through the MJX-style functional API
(myosuite_amd/mjx_api.py: make / reset / step over a State).  The reference runs every id on two implementations
("warp", "jax"); here both slots are this engine ("hip").  Results go to `mjx_benchmark_results.npy` (the reference's file
name and layout: {f"{env_name}_{impl}": [seconds per E]}) and, with --json, to a JSON file with env-steps/s next to them.

    python benchmarks/mjx_benchmark.py [--env MjxHandReachRandom-v0 ...] [--eager] [--json gpurun_out/mjx_benchmark.json]
"""
import argparse
import json
import os
import sys
import timeit

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from myosuite_amd import engine as E
from myosuite_amd import mjx_api

ENV_NAMES = ["MjxElbowPoseRandom-v0", "MjxFingerPoseRandom-v0", "MjxHandReachRandom-v0"]       # mjx_benchmark.py:54
NUM_ENVS = [64, 512, 1024, 2048, 4096, 8192]                                                     # mjx_benchmark.py:22


def measure_num_env_simulation_steps(seed=0, loop_iterations=16, env_name="MjxElbowPoseRandom-v0", impl="hip", graph=True):
    """Measure how the number of envs influences execution time (total number of steps): mjx_benchmark.py:11-50.

    graph=True: the 16-step loop is captured once into a HIP graph and replayed -- the counterpart of the reference's jitted
    ``jax.lax.scan`` over 16 steps (mjx_benchmark.py:24-38); actions are drawn inside the graph (torch's graph-safe Philox), as
    the reference draws them inside the scan.  graph=False launches every step eagerly."""
    res = []
    for e in NUM_ENVS:
        env = mjx_api.make(env_name, config_overrides={"impl": impl}, num_envs=e, seed=seed)
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

        print(f"Testing env num: {e}")
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
              f"({8192 * loop_iterations / np.mean(results):.3e} env-steps/s).")
        res.append(float(np.mean(results)))
        del env
    print("[" + ", ".join(str(r) for r in res) + "]")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", action="append", default=None, help="subset of the reference's three env ids (default: all)")
    ap.add_argument("--eager", action="store_true", help="launch every step instead of replaying a captured 16-step HIP graph")
    ap.add_argument("--json", default=None, help="also write {key: {seconds, env_steps_per_s}} here")
    ap.add_argument("--out", default="mjx_benchmark_results.npy")
    a = ap.parse_args()
    results = {}
    for env_name in (a.env or ENV_NAMES):
        for impl in ["hip"]:       # the reference's two slots ("warp", "jax") are one engine here
            print(f"MyoSuite env {env_name} -- Testing implementation: {impl}")
            results[f"{env_name}_{impl}"] = measure_num_env_simulation_steps(seed=0, env_name=env_name, impl=impl, graph=not a.eager)
    print(results)
    np.save(a.out, results, allow_pickle=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        js = {"protocol": "benchmarks/mjx_benchmark.py of the reference: 16-step loop, 8192*16 total env-steps per measurement, mean of 3 repeats; "
                          + ("16-step loop replayed as one HIP graph" if not a.eager else "eager launches"),
              "num_envs": NUM_ENVS,
              "results": {k: {"seconds": v, "env_steps_per_s": [8192 * 16 / s for s in v]} for k, v in results.items()}}
        json.dump(js, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
