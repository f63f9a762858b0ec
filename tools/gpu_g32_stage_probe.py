#!/usr/bin/env python
"""Stage profile of the self-colliding hand on the 32-lane general-row kernel (two envs per wave; row bound cut to 32: rows beyond
are dropped and flagged, under random actions that is ~never) next to the 64-lane one.  Needs a -DMM_STAGE_PROF=1 library."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from myosuite_amd.envs import registry
import bench

synth._CACHE["hand_contact_j32"] = synth.compile_spec("hand_contact", edit=lambda s: setattr(s, "njmax", 32))
for model, n in (("hand_contact", 4096), ("hand_contact_j32", 4096)):
    env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=0, model=model)
    env.rollout_setup(action_seed=0)
    for s in range(6): env.rollout_step(None, stream_id=s)
    torch.cuda.synchronize()
    K = 40
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for s in range(K): env.rollout_step(None, stream_id=6 + s, events=evs[s])
    torch.cuda.synchronize()
    km = float(np.median([a.elapsed_time(b) for a, b in evs]))
    a = torch.rand(n, env.cm.nu, device="cuda")
    try:
        pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
    except Exception as exc:
        pf = {"error": repr(exc)}
    nf = env.frame_skip + 1
    print(f"{model:18s} G={env.hm.launch_lanes(n):2d} kernel {km:.4f} ms  {n / km / 1e3:.3f} M/s status_or {bench.status_or(env.state.status)}  " +
          " ".join(f"{k}:{(v // nf) if isinstance(v, int) else v}" for k, v in pf.items() if v), flush=True)
    print("   launch", env.hm.launch_info(n), flush=True)
