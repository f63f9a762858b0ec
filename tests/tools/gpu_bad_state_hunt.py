"""Find the first bad-state reset (status bit 1) in a random rollout of an env id and replay that env-step in the oracle from the
state before it:  python tests/tools/gpu_bad_state_hunt.py --env myoFatiHandPenTwirlRandom-v0 [--out gpurun_out/bad_state_hunt.json]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch                          # noqa: E402
from myosuite_amd import engine as E               # noqa: E402
from myosuite_amd.envs import registry             # noqa: E402
from oracle import oracle as O                     # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--env", default="myoFatiHandPenTwirlRandom-v0"); ap.add_argument("--out", default=None)
ap.add_argument("--steps", type=int, default=400); ap.add_argument("--num-envs", type=int, default=64)
a = ap.parse_args()
O.build()
env = registry.make(a.env, num_envs=a.num_envs, seed=1)
env.reset(seed=1)
n = a.num_envs
act = torch.empty(n, env.action_space.shape[0], device="cuda")
events = []
for s in range(a.steps):
    E.uniform(act, 7, s)
    if s % 3 == 0:
        act.mul_(2).sub_(1)
    pre = {k: (v.clone() if v is not None else None) for k, v in env.get_env_state().items()}
    pre_status = env.state.status.clone()
    gs = env.state.geom_size_env.clone() if getattr(env.state, "geom_size_env", None) is not None else None
    gt = env.state.geom_type_env.clone() if getattr(env.state, "geom_type_env", None) is not None else None
    env_autoreset = env.autoreset
    env.autoreset = False                         # look at the status before the reset clears it
    o, r, te, tr, info = env.step(act)
    st = env.state.status.cpu().numpy()
    new = np.nonzero((st & 1) & ~(pre_status.cpu().numpy() & 1))[0]
    for e in new[:2]:
        ctrl = env.last_ctrl[e].cpu().numpy().astype(np.float64)
        om = O.OracleModel(env.cm)
        d = O.OracleData(om)
        d.qpos[:] = pre["qpos"][e].cpu().numpy(); d.qvel[:] = pre["qvel"][e].cpu().numpy()
        if env.cm.na: d.act[:] = pre["act"][e].cpu().numpy()
        d.qacc_warmstart[:] = pre["qacc_warmstart"][e].cpu().numpy()
        if gs is not None and env.state._c.geom_env_id >= 0:
            d.set_geom_size(int(env.state._c.geom_env_id), gs[e].cpu().numpy(), int(gt[e]) if gt is not None else -1)
        d.ctrl[:] = ctrl
        worst = 0.0; warn_at = -1
        for k in range(env.frame_skip):
            d.step(1)
            worst = max(worst, float(np.abs(d.qacc).max()))
            if d.warn & 1 and warn_at < 0: warn_at = k
        events.append({"step": s, "env": int(e), "oracle_warn": int(d.warn), "oracle_bad_at_substep": warn_at, "oracle_max_abs_qacc": worst,
                       "pre_max_abs_qvel": float(pre["qvel"][e].abs().max()), "pre_qpos_tail7": pre["qpos"][e][-7:].cpu().numpy().round(4).tolist(),
                       "ctrl_max": float(np.abs(ctrl).max()), "gpu_status": int(st[e])})
    env.autoreset = env_autoreset
    if env.autoreset:
        env.reset(mask=(env.done | env.truncated))
    if len(events) >= 4:
        break
out = {"env": a.env, "steps_run": s + 1, "events": events}
print(json.dumps(out, indent=1))
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)
