"""Compact perf report of the four BASELINE workloads (kernel ms, env-steps/s) + in-kernel stage profile.
python tools/gpu_perf.py [hand|elbow|leg|reorient ...] [--lib path/to/variant.so ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
W = {"elbow": ("myoElbowPose1D6MRandom-v0", 4096), "hand": ("myoHandPoseRandom-v0", 4096), "leg": ("myoFatiLegWalk-v0", 1024),
     "reorient": ("myoHandReorient100-v0", 2048), "contact": ("myoHandPoseRandom-v0", 4096), "legi": ("myoFatiLegWalk-v0", 1024),
     "dense": ("myoHandReorient100-v0", 2048),       # model="hand_dense": 189 candidate pairs
     "hand64": ("myoHandPoseRandom-v0", 4096), "elbow64": ("myoElbowPose1D6MRandom-v0", 4096)}      # precision mode (fp64 state rows)
names = [a for a in sys.argv[1:] if a in W] or ["elbow", "hand", "leg", "reorient"]
for nm in names:
    env_id, n = W[nm]
    env = registry.make(env_id, num_envs=n, seed=0, **({"model": "hand_contact"} if nm == "contact" else {"model": "hand_dense"} if nm == "dense" else ({"model": "leg_implicit"} if nm == "legi" else
                                                                ({"precision": "f64_state"} if nm.endswith("64") else {}))))
    env.rollout_setup(action_seed=0)
    for s in range(int(os.environ.get("PERF_WARM_STEPS", "6"))): env.rollout_step(None, stream_id=s)     # (PERF_WARM_STEPS=200: the steady mix of episode phases bench.py times)
    torch.cuda.synchronize()
    K = 48
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    t0 = time.perf_counter()
    for s in range(K): env.rollout_step(None, stream_id=1000 + s, events=evs[s])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    km = float(np.median([a.elapsed_time(b) for a, b in evs]))
    a = torch.rand(n, env.cm.nu, device="cuda")
    pf = E.profile_stages(lambda: E.env_step(env.hm, env.state, a, env._task))
    nf = env.frame_skip + 1
    print(f"{nm:9s} G={env.hm.launch_lanes(n):2d} {n * K / dt / 1e6:7.3f} M env-steps/s  kernel {km:.4f} ms   " +
          " ".join(f"{k}:{v // nf}" for k, v in pf.items() if v))
