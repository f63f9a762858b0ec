"""Newton iteration / active-row statistics of the final forward pass along a random-action rollout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
for env_id, n in (("myoHandPoseRandom-v0", 4096), ("myoElbowPose1D6MRandom-v0", 4096), ("myoFatiLegWalk-v0", 1024), ("myoHandReorient100-v0", 2048)):
    if len(sys.argv) > 1 and not any(a in env_id for a in sys.argv[1:]): continue
    env = registry.make(env_id, num_envs=n, seed=0)
    env.reset(seed=0)
    dv = E.Derived(env.hm, n, ["nefc", "solver_niter"])
    hist = np.zeros(12); nefc = []
    for s in range(40):
        a = torch.rand(n, env.cm.nu, device="cuda")
        E.env_step(env.hm, env.state, a, env._task, dv)
        if s >= 10:
            it = dv.t["solver_niter"].cpu().numpy(); hist += np.bincount(np.minimum(it, 11), minlength=12)
            nefc.append(dv.t["nefc"].float().mean().item())
    print(env_id, "niter hist", (hist / hist.sum()).round(3), "mean", (hist * np.arange(12)).sum() / hist.sum(), "mean nefc", np.mean(nefc))
