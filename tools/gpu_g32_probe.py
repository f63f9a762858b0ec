#!/usr/bin/env python
"""What would the hand-family general-row models gain from two envs per wave?  The row budget (one constraint row per lane) forces
them onto 64 lanes per env; with the contact budget cut to nconmax = 2 they fit the existing 32-lane general-row kernels (surplus
rows are dropped: status bit 8, so this is a THROUGHPUT probe, not a valid simulation)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.model import synth
from myosuite_amd.envs import registry
import bench

def probe(env_id, n, model, steps=40):
    env = registry.make(env_id, num_envs=n, seed=0, model=model)
    env.rollout_setup(action_seed=0)
    for s in range(6):
        env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s in range(steps):
        env.rollout_step(None, stream_id=6 + s, events=evs[s])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    km = sum(a.elapsed_time(b) for a, b in evs) / steps
    return {"model": model, "lanes": env.hm.launch_lanes(n), "env_steps_per_s": n * steps / dt, "kernel_ms": km, "status_or": bench.status_or(env.state.status)}

for base, env_id, n in (("hand_contact", "myoHandPoseRandom-v0", 4096), ("hand_reorient", "myoHandReorient100-v0", 2048)):
    synth._CACHE[base + "_c2"] = synth.compile_spec(base, edit=lambda s: setattr(s, "nconmax", 2))
    for m in (base, base + "_c2"):
        print(json.dumps(probe(env_id, n, m)), flush=True)
