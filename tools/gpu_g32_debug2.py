import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from myosuite_amd.envs import registry
case, n = sys.argv[1], int(sys.argv[2])
synth._CACHE["hand_contact_c2"] = synth.compile_spec("hand_contact", edit=lambda s: setattr(s, "nconmax", 2))
env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=0, model="hand_contact_c2", autoreset=("auto" in case))
print(case, n, "lanes", env.hm.launch_lanes(n), flush=True)
a = torch.rand(n, env.cm.nu, device="cuda")
if "envstep" in case:
    for k in range(10):
        env.step(a); torch.cuda.synchronize()
    print(case, "env.step ok", flush=True)
if "rollout" in case:
    env.rollout_setup(action_seed=0)
    for k in range(10):
        env.rollout_step(None, stream_id=k); torch.cuda.synchronize()
    print(case, "rollout ok", flush=True)
