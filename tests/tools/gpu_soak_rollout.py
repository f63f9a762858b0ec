"""Soak of the fused rollout launch at the bench batches: N env-steps of random actions per workload (episodes end and re-arm inside
the launch), status bits / finiteness / episode statistics checked every 500 steps.  Writes gpurun_out/soak_rollout.json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry

N = int(os.environ.get("SOAK_STEPS", 3000))
W = [("myoHandPoseRandom-v0", 4096, {}), ("myoElbowPose1D6MRandom-v0", 4096, {}), ("myoHandReachRandom-v0", 4096, {}),
     ("myoHandPoseRandom-v0", 4096, {"model": "hand_contact"}), ("myoHandReorient100-v0", 2048, {}), ("myoFatiLegWalk-v0", 1024, {}),
     ("myoFatiLegWalk-v0", 1024, {"model": "leg_implicit"}), ("myoHandKeyTurnRandom-v0", 2048, {}), ("myoHandPenTwirlRandom-v0", 2048, {}),
     ("myoHandPoseRandom-v0", 4096, {"precision": "f64_state"}), ("myoHandReorient100-v0", 1024, {"precision": "f64_state"}),
     ("myoFatiLegWalk-v0", 512, {"precision": "f64_state"})]
out = {}
for env_id, n, kw in W:
    key = f"{env_id}@{n}" + "".join(f"|{k}={v}" for k, v in kw.items())
    env = registry.make(env_id, num_envs=n, seed=7, **kw)
    stats = env.rollout_setup(action_seed=11)
    finite, ever = True, 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(N):
        obs, rw, mask = env.rollout_step(None, stream_id=s)
        if s % 500 == 499:
            finite &= bool(torch.isfinite(obs).all()) and bool(torch.isfinite(rw).all()) and bool(torch.isfinite(env.state.qpos).all())
            ever |= int(env.state.status.max())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = env.state.status
    out[key] = {"envs": n, "env_steps": N, "finite": finite, "status_or": int(ever) | int(st.max()),
                "bad_state_resets_flagged": int((st & 1).sum()), "solver_cap_flagged": int(((st >> 2) & 1).sum()),
                "row_overflow_flagged": int(((st >> 3) & 1).sum()), "partner_wave_timeout_flagged": int(((st >> 4) & 1).sum()),
                "episodes_per_env": float(stats[:, 1].sum() / max(1.0, float(n)) / max(1, env.max_episode_steps)) if stats is not None else None,
                "mean_return": float(stats[:, 0].mean()), "env_steps_per_s_incl_checks": round(n * N / dt)}
    print(key, out[key], flush=True)
    del env
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/soak_rollout.json", "w"), indent=1)
