"""One-off soak of EVERY registered env id: 64 envs x 300 env-steps of random actions with auto-reset; records non-finite outputs and
status bits (1 bad-state reset, 4 solver cap, 8 row overflow, 16 lost partner wave, 32 bad control) per id.
    python tests/tools/gpu_all_ids_soak.py [--out gpurun_out/all_ids_soak.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch                                       # noqa: E402
from myosuite_amd import engine as E               # noqa: E402
from myosuite_amd.envs import registry             # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--num-envs", type=int, default=64)
a = ap.parse_args()
t0 = time.time()
rows = {}
for env_id in sorted(registry._SPECS):
    try:
        env = registry.make(env_id, num_envs=a.num_envs, seed=1)
        env.reset(seed=1)
        act = torch.empty(a.num_envs, env.action_space.shape[0], device="cuda")
        bits, finite, ndone, rsum = 0, True, 0, 0.0
        for s in range(a.steps):
            E.uniform(act, 7, s)
            if s % 3 == 0:
                act.mul_(2).sub_(1)                # the [-1, 1] end of the action box too
            o, r, te, tr, _ = env.step(act)
            bits |= int(env.state.status.max().item()) if s % 25 == 24 or s == a.steps - 1 else 0
            if s % 25 == 24 or s == a.steps - 1:
                finite = finite and bool(torch.isfinite(o).all() and torch.isfinite(r).all())
            ndone += int((te | tr).sum()); rsum += float(r.mean())
        sticky = 0
        st = env.state.status.cpu().numpy()
        for b in (1, 4, 8, 16, 32):
            if (st & b).any():
                sticky |= b
        rows[env_id] = {"finite": finite, "status_or_sampled": bits, "status_or_final": sticky, "episodes_ended": ndone,
                        "mean_reward_per_step": rsum / a.steps}
        del env
    except Exception as e:                          # noqa: BLE001
        rows[env_id] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
bad = {k: v for k, v in rows.items() if v.get("error") or not v.get("finite", False) or (v.get("status_or_sampled", 0) | v.get("status_or_final", 0)) & (1 | 16 | 32)}
out = {"ids": len(rows), "steps": a.steps, "num_envs": a.num_envs, "elapsed_s": round(time.time() - t0, 1),
       "ids_with_errors_or_nonfinite_or_bits_1_16_32": bad,
       "ids_with_solver_cap_or_row_overflow": sorted(k for k, v in rows.items() if (v.get("status_or_sampled", 0) | v.get("status_or_final", 0)) & 12), "rows": rows}
print(json.dumps({k: v for k, v in out.items() if k != "rows"}, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
