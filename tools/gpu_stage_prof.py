import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
for env_id, n in (("myoElbowPose1D6MRandom-v0", 4096), ("myoHandPoseRandom-v0", 4096), ("myoLegWalk-v0", 1024), ("myoHandReorient100-v0", 2048)):
    env = registry.make(env_id, num_envs=n, seed=0)
    a = torch.rand(n, env.cm.nu, device="cuda")
    for _ in range(3): env.step(a)
    pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
    tot = pf["total"]
    nf = env.frame_skip + 1
    print(env_id, "lanes", env.hm.launch_lanes(n), "lds/env", env.hm.info(E.INFO_LDS_PER_ENV), {k: (v // nf, round(100.0 * v / tot, 1)) for k, v in pf.items()})
    if "Reorient" in env_id:
        print("  env 0 object type", int(env.geom_type[0]), "contacts/rows: nefc via derived n/a")
