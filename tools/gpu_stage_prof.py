import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
for env_id, n in (("myoHandPoseRandom-v0", 4096), ("myoLegWalk-v0", 1024)):
    env = registry.make(env_id, num_envs=n, seed=0)
    a = torch.rand(n, env.cm.nu, device="cuda")
    for _ in range(3): env.step(a)
    pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
    tot = pf["total"]
    print(env_id, "lanes", env.hm.info(E.INFO_LANES), {k: (v // 11, round(100.0 * v / tot, 1)) for k, v in pf.items()})
