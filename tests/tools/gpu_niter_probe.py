"""Newton iterations per solve at the bench batches (steady state of a random-action rollout): mean / histogram of solver_niter"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
W = {"hand": ("myoHandPoseRandom-v0", 4096, {}), "contact": ("myoHandPoseRandom-v0", 4096, {"model": "hand_contact"}), "leg": ("myoFatiLegWalk-v0", 1024, {}),
     "legi": ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}), "reorient": ("myoHandReorient100-v0", 2048, {})}
for nm, (env_id, n, kw) in W.items():
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    env.rollout_setup(action_seed=0)
    hist = np.zeros(12)
    rows = []
    d_ = E.Derived(env.hm, n, ["nefc", "solver_niter"])
    for s in range(120):
        env.rollout_step(None, stream_id=s)
        if s >= 60 and s % 6 == 0:
            E.forward(env.hm, env.state, env.last_ctrl.clone(), d_)
            it = d_["solver_niter"].cpu().numpy().astype(int)
            hist += np.bincount(np.clip(it, 0, 11), minlength=12)
            rows.append(float(d_["nefc"].float().mean()))
    p = hist / hist.sum()
    print(f"{nm:9s} mean iterations {float((p * np.arange(12)).sum()):.2f}  share by count 0..6+: " + " ".join(f"{x:.2f}" for x in list(p[:6]) + [p[6:].sum()]) + f"  mean rows {np.mean(rows):.1f}")
