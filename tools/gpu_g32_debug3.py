import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myosuite_amd.model import synth
from myosuite_amd.envs import registry
mode, n, model = sys.argv[1], int(sys.argv[2]), sys.argv[3]
synth._CACHE["hand_contact_c2"] = synth.compile_spec("hand_contact", edit=lambda s: setattr(s, "nconmax", 2))
env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=0, model=model)
print(mode, model, n, "lanes", env.hm.launch_lanes(n), flush=True)
env.rollout_setup(action_seed=0)
for k in range(12):
    env.rollout_step(None, stream_id=k)
    if mode == "sync":
        torch.cuda.synchronize()
torch.cuda.synchronize(); print(mode, "done status", int(env.state.status.max()), flush=True)
