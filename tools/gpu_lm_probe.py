#!/usr/bin/env python
"""What is staging the model tables in LDS worth to a general-row kernel?  Same batch, same launch shape (two-wave launches off),
lds_model = 0 (tables through L2) vs 2 (tables in LDS), at a batch where both fit the same number of waves per CU."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry

def probe(env_id, n, lm, steps=30, **kw):
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    env.hm.set_option("lds_model", lm)
    env.rollout_setup(action_seed=0)
    for s in range(5):
        env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for s in range(steps):
        env.rollout_step(None, stream_id=5 + s, events=evs[s])
    torch.cuda.synchronize()
    km = sum(a.elapsed_time(b) for a, b in evs) / steps
    return {"env": env_id, "n": n, "lds_model": lm, "kernel_ms": round(km, 4), "lds_per_env": env.hm.info(E.INFO_LDS_PER_ENV), "model_words": env.hm.info(E.INFO_MODEL_WORDS), **kw}

for env_id, n, kw in (("myoHandReorient100-v0", 1024, {}), ("myoHandReorient100-v0", 2048, {}), ("myoFatiLegWalk-v0", 1024, {}),
                      ("myoHandPoseRandom-v0", 2048, {"model": "hand_contact"}), ("myoHandPoseRandom-v0", 4096, {"model": "hand_contact"})):
    for lm in (0, 2):
        try:
            print(json.dumps(probe(env_id, n, lm, **kw)), flush=True)
        except Exception as ex:
            print(env_id, n, lm, "failed:", ex, flush=True)
