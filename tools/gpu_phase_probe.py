"""Kernel time of the fused env-step launch by episode phase: per-step HIP-event durations of a fresh rollout, binned by step index.
The light launches (leg: 1024 env waves) are latency-bound chains whose length follows the constraint count of the state -- a
protocol that times steps 10..42 of the episode and one that times steps 100..300 measure different physics."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
W = {"leg": ("myoFatiLegWalk-v0", 1024, {}), "legi": ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}), "reorient": ("myoHandReorient100-v0", 2048, {}),
     "contact": ("myoHandPoseRandom-v0", 4096, {"model": "hand_contact"}), "hand": ("myoHandPoseRandom-v0", 4096, {})}
res = {}
for nm, (env_id, n, kw) in W.items():
    env = registry.make(env_id, num_envs=n, seed=0, **kw)
    env.rollout_setup(action_seed=0)
    K = 400
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    nefc = []
    d_ = E.Derived(env.hm, n, ["nefc"])
    for s in range(K):
        env.rollout_step(None, stream_id=s, events=evs[s])
        if s % 50 == 25:
            torch.cuda.synchronize()
            E.forward(env.hm, env.state, env.last_ctrl.clone(), d_)
            nefc.append(float(d_["nefc"].float().mean()))
    torch.cuda.synchronize()
    km = np.array([a.elapsed_time(b) for a, b in evs])
    bins = [float(np.median(km[i:i + 50])) for i in range(0, K, 50)]
    res[nm] = {"kernel_ms_median_by_50_steps": bins, "mean_rows_at_step_25_75_etc": nefc, "episode_steps": int(env.max_episode_steps)}
    print(nm, "episode", env.max_episode_steps, "kernel ms by 50-step bin:", " ".join(f"{b:.3f}" for b in bins), "| mean rows:", " ".join(f"{x:.1f}" for x in nefc), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/phase_probe.json", "w"), indent=1)
