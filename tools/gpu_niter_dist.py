"""Distribution of Newton iterations / constraint rows over the envs of a batch early in an episode and in the steady mix of episode
phases (what decides the launch time is the SLOWEST env: every wave is resident at once).  python tools/gpu_niter_dist.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
for env_id, n, kw in (("myoFatiLegWalk-v0", 1024, {}), ("myoHandReorient100-v0", 2048, {}), ("myoHandPoseRandom-v0", 4096, {})):
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    env.rollout_setup(action_seed=0)
    s = 0
    for warm in (6, 250):
        while s < warm:
            env.rollout_step(None, stream_id=s); s += 1
        d_ = E.Derived(env.hm, n, ["nefc", "solver_niter"])
        E.forward(env.hm, env.state, env.last_ctrl.clone(), d_)
        torch.cuda.synchronize()
        it, ne = d_["solver_niter"].cpu().numpy(), d_["nefc"].cpu().numpy()
        print(f"{env_id} after {warm} steps: niter hist {np.bincount(it, minlength=8)[:12].tolist()} max {it.max()} mean {it.mean():.2f}; nefc median {int(np.median(ne))} max {ne.max()}; step_count mean {float(env.step_count.float().mean()):.1f}")
