"""(1) Soak: every env family at full batch, random actions, 300 env-steps: finite obs/reward, status bits, resets.
(2) Contact-model parity: teacher-forced per-env-step error of the leg-walk and reorient envs against the fp64 oracle
over 100 env-steps (free-running comparison is not meaningful across contact onsets).
Writes gpurun_out/soak_contact_parity.json."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
from myosuite_amd.model import synth
from oracle import env_oracle as EO

out = {"soak": {}, "teacher_forced": {}}
for env_id, n in (("myoElbowPose1D6MRandom-v0", 4096), ("myoHandPoseRandom-v0", 4096), ("myoHandReachRandom-v0", 4096),
                  ("myoHandReorient100-v0", 2048), ("myoFatiLegWalk-v0", 1024), ("myoHandPenTwirlRandom-v0", 2048),
                  ("myoHandObjHoldRandom-v0", 2048), ("myoHandKeyTurnRandom-v0", 2048), ("myoTorsoPoseFixed-v0", 1024),
                  ("myoFingerReachRandom-v0", 4096), ("motorFingerPoseRandom-v0", 4096), ("myoElbowPose1D6MExoRandom-v0", 4096),
                  ("myoLegStandRandom-v0", 1024)):
    env = registry.make(env_id, num_envs=n, seed=3)
    env.reset(seed=3)
    a = torch.empty(n, env.cm.nu, device="cuda")
    ndone = 0; finite = True
    for s in range(300):
        E.uniform(a, 5, s)
        obs, r, term, trunc, info = env.step(a)
        ndone += int((term | trunc).sum())
        finite &= bool(torch.isfinite(obs).all()) and bool(torch.isfinite(r).all())
    st = env.state.status
    out["soak"][env_id] = dict(envs=n, steps=300, finite=finite, episodes_finished=ndone,
                               bad_state_resets=int((st & 1).sum()), row_overflow=int(((st >> 3) & 1).sum()),
                               solver_cap=int(((st >> 2) & 1).sum()))
    torch.cuda.synchronize(); t0 = __import__("time").perf_counter()
    for s in range(20):
        env.step(a)
    torch.cuda.synchronize()
    out["soak"][env_id]["env_steps_per_s"] = round(20 * n / (__import__("time").perf_counter() - t0))
    print(env_id, out["soak"][env_id])
    del env

def forced(env_id, make_oracle, reset_oracle, nsteps=100, n=8, scale=0.6):
    cm = synth.get_model(registry.spec(env_id)["kwargs"]["model"])
    env = registry.make(env_id, num_envs=n, seed=1, autoreset=False)
    env.reset(seed=1)
    orc = [make_oracle(cm) for _ in range(n)]
    for e in range(n):
        reset_oracle(env, orc[e], e)
    a = torch.empty(n, cm.nu, device="cuda")
    eq, ev = [], []
    for s in range(nsteps):
        st = env.get_env_state()
        for e in range(n):
            d = orc[e].d
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                v = getattr(d, k).astype(np.float32); getattr(d, k)[:] = v; st[k][e] = torch.from_numpy(v)
        env.set_env_state(st)
        E.uniform(a, 2, s)
        act = (scale * a).contiguous()
        env.step(act)
        an = act.cpu().numpy()
        for e in range(n):
            orc[e].step(an[e].astype(np.float64))
        q = env.state.qpos.cpu().numpy(); v = env.state.qvel.cpu().numpy()
        eq.append(np.abs(q - np.stack([o.d.qpos for o in orc])).max(axis=1)); ev.append(np.abs(v - np.stack([o.d.qvel for o in orc])).max(axis=1))
    eq, ev = np.array(eq), np.array(ev)
    return dict(env_steps=nsteps, envs=n, qpos_abs_err_median=float(np.median(eq)), qpos_abs_err_p99=float(np.quantile(eq, 0.99)),
                qpos_abs_err_max=float(eq.max()), qvel_abs_err_median=float(np.median(ev)), qvel_abs_err_max=float(ev.max()))

def reset_walk(env, w, e):
    cm = w.cm
    w.reset(cm.key_qpos[2], cm.key_qvel[2])
def reset_reor(env, w, e):
    w.reset(env.geom_size[e].cpu().numpy().astype(np.float64), float(env.axis_half[e]), env.des_rot[e].cpu().numpy().astype(np.float64), int(env.geom_type[e]))
out["teacher_forced"]["myoLegWalk-v0"] = forced("myoLegWalk-v0", EO.WalkEnvOracle, reset_walk)
print(out["teacher_forced"]["myoLegWalk-v0"])
out["teacher_forced"]["myoHandReorient100-v0"] = forced("myoHandReorient100-v0", EO.ReorientEnvOracle, reset_reor, scale=0.8)
print(out["teacher_forced"]["myoHandReorient100-v0"])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/soak_contact_parity.json", "w"), indent=1)
