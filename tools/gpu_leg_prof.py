import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
env = registry.make("myoLegWalk-v0", num_envs=1024, seed=0)
a = torch.rand(1024, env.cm.nu, device="cuda")
for _ in range(3): env.step(a)
pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
tot = pf["total"]
print({k: (v, round(100.0 * v / tot, 1)) for k, v in pf.items()})
print("niter", "lanes", env.hm.info(E.INFO_LANES), "lds/env", env.hm.info(E.INFO_LDS_PER_ENV))
