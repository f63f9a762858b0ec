"""Where the fp32 kernel loses accuracy against the fp64 oracle (hand model): per-stage relative error of one forward pass and
the 1000-step free-running divergence, with the kernel's internal frame at the world origin (origin_shift = 0) and centred
on the model (origin_shift = 1, the default).   python tests/tools/fp32_error_study.py  ->  gpurun_out/fp32_error_study.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from myosuite_amd import engine as E
from myosuite_amd.model import synth
from oracle import env_oracle as EO
from oracle import oracle as O

OMAP = {"tenlen": "ten_length", "tenvel": "ten_velocity", "actfrc": "actuator_force", "actdot": "act_dot", "bias": "qfrc_bias",
        "smooth": "qfrc_smooth", "qaccsm": "qacc_smooth"}
NAMES = ["xpos", "xipos", "cdof", "cvel", "tenlen", "tenvel", "actfrc", "bias", "smooth", "qaccsm", "qacc"]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(1e-9, np.abs(b).max()))


def stage_errors(cm, hm, nenv=64):
    om = O.OracleModel(cm)
    rng = np.random.default_rng(0)
    lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
    qpos = (lo + (hi - lo) * rng.random((nenv, cm.nq))).astype(np.float32)
    qvel = (rng.standard_normal((nenv, cm.nv)) * 2).astype(np.float32)
    act = rng.random((nenv, cm.na)).astype(np.float32); ctrl = rng.random((nenv, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, nenv)
    st.qpos.copy_(torch.from_numpy(qpos)); st.qvel.copy_(torch.from_numpy(qvel)); st.act.copy_(torch.from_numpy(act))
    dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda()).cpu().numpy()
    worst = {n: 0.0 for n in NAMES + ["tenJ", "M"]}
    for e in range(nenv):
        d = O.OracleData(om)
        d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]
        d.forward()
        for n in NAMES:
            ref = getattr(d, OMAP.get(n, n)).ravel()
            worst[n] = max(worst[n], rel(dump[e, hm.layout(n):hm.layout(n) + ref.size], ref))
        M = dump[e, hm.layout("M"):hm.layout("M") + cm.nv * cm.nv].reshape(cm.nv, cm.nv)
        worst["M"] = max(worst["M"], rel(M, d.full_M()))
        # sparse tendon Jacobian in the blob's pattern
        adr = cm.arrays["TENJ_ADR"]; dof = cm.arrays["TENJ_DOF"]
        tj = dump[e, hm.layout("tenj"):hm.layout("tenj") + len(dof)]
        J = d.ten_J
        ref = np.array([J[t, dof[k]] for t in range(cm.ntendon) for k in range(adr[t], adr[t + 1])])
        worst["tenJ"] = max(worst["tenJ"], rel(tj, ref))
    return worst


def divergence(cm, hm, nenv=16, nsteps=100):
    om = O.OracleModel(cm)
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    q0 = np.stack([(lo + (hi - lo) * EO.pose_reset_draws(cm.nq, e, 0, 0)[0]).astype(np.float32) for e in range(nenv)])
    st = E.BatchState(hm, nenv); st.qpos.copy_(torch.from_numpy(q0))
    ds = []
    for e in range(nenv):
        d = O.OracleData(om); d.qpos[:] = q0[e]; ds.append(d)
    a = torch.empty(nenv, cm.nu, device="cuda")
    relq = []
    for s in range(nsteps):
        E.uniform(a, 0, s)
        ctrl = (1.0 / (1.0 + torch.exp(-5.0 * (a - 0.5)))).contiguous()
        E.step(hm, st, ctrl, 10)
        c = ctrl.cpu().numpy()
        for e, d in enumerate(ds):
            d.ctrl[:] = c[e]; d.step(10)
        oq = np.stack([d.qpos for d in ds]); gq = st.qpos.cpu().numpy()
        relq.append(float(np.abs(gq - oq).max() / max(1.0, np.abs(oq).max())))
    return relq


out = {}
for name in ("hand",):
    cm = synth.get_model(name)
    for lanes in (32, 64):
        for shift in (0, 1):
            hm = E.HipModel(cm, lanes_per_env=lanes)
            hm.set_option("origin_shift", shift)
            se = stage_errors(cm, hm)
            dv = divergence(cm, hm)
            key = f"{name}_G{lanes}_shift{shift}"
            out[key] = {"stage_rel_err": se, "div_at_100_300_1000": [dv[9], dv[29], dv[99]], "div_max_over_run": max(dv),
                        "div_argmax_env_step": int(np.argmax(dv))}
            print(key, "stages:", {k: f"{v:.1e}" for k, v in se.items()})
            print(key, "divergence 100/300/1000: %.2e %.2e %.2e  max %.2e @ env-step %d" % (dv[9], dv[29], dv[99], max(dv), int(np.argmax(dv))))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/fp32_error_study.json", "w"), indent=1)
