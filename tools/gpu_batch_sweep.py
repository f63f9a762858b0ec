"""env-steps/s of the fused rollout launch against the batch size (envs per GPU), one GPU: where each workload saturates the chip.
Writes gpurun_out/batch_sweep.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.envs import registry
W = {"hand": ("myoHandPoseRandom-v0", {}), "elbow": ("myoElbowPose1D6MRandom-v0", {}), "hand_contact": ("myoHandPoseRandom-v0", {"model": "hand_contact"}),
     "reorient": ("myoHandReorient100-v0", {}), "fati-leg": ("myoFatiLegWalk-v0", {})}
out = {}
for nm, (env_id, kw) in W.items():
    out[nm] = {}
    for n in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536):
        env = registry.make(env_id, num_envs=n, seed=0, **kw)
        env.rollout_setup(action_seed=0)
        for s in range(4): env.rollout_step(None, stream_id=s)
        K = max(8, min(64, int(40.0 / (0.6 * max(1.0, n / 4096)))))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for s in range(K): env.rollout_step(None, stream_id=4 + s)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[nm][n] = {"env_steps_per_s": n * K / dt, "ms_per_step": 1e3 * dt / K, "lanes_per_env": env.hm.launch_lanes(n)}
        del env
    print(nm, " ".join(f"{n}:{v['env_steps_per_s'] / 1e6:.2f}M(G{v['lanes_per_env']})" for n, v in out[nm].items()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/batch_sweep.json", "w"), indent=1)
