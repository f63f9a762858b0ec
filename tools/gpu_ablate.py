#!/usr/bin/env python
"""Ablation timing of the hand kernel's solver knobs (results are WRONG in the ablated runs: this only measures what a part costs
in kernel time, latency effects included -- the in-kernel stage timers misattribute across outstanding LDS operations)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd.envs import registry

def timeit(model, n=4096, steps=40):
    env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=0, model=model)
    env.rollout_setup(action_seed=0)
    for s in range(6):
        env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s in range(steps):
        env.rollout_step(None, stream_id=6 + s, events=evs[s])
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))

def variant(name, **kw):
    def edit(s):
        for k, v in kw.items():
            setattr(s, k, v)
    synth._CACHE[name] = synth.compile_spec("hand", edit=edit)
    return name

base = timeit("hand")
print("base", round(base, 4), flush=True)
for nm, kw in (("ls2", dict(ls_iterations=2)), ("ls1", dict(ls_iterations=1)), ("it1", dict(iterations=1)), ("it1_ls1", dict(iterations=1, ls_iterations=1)), ("it0", dict(iterations=0))):
    try:
        t = timeit(variant("hand_" + nm, **kw))
        print(nm, round(t, 4), f"delta {100 * (base - t) / base:.1f} %", flush=True)
    except Exception as ex:
        print(nm, "failed", ex, flush=True)
print("base again", round(timeit("hand"), 4))
