"""GPU bring-up diagnostic: compare every LDS workspace buffer of the HIP engine's forward pass
against the fp64 oracle on random states, then compare multi-step rollouts.  Run on a GPU box:
    python tools/gpu_debug.py [elbow|hand] [lanes]
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0, 0.0
    return float(np.abs(a - b).max()), float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def main(name, lanes):
    cm = synth.get_model(name)
    om = O.OracleModel(cm)
    hm = E.HipModel(cm, lanes_per_env=lanes)
    print(f"== {name}: lanes/env={hm.info(E.INFO_LANES)} lds/env={hm.info(E.INFO_LDS_PER_ENV)}B")
    nenv = 37
    rng = np.random.default_rng(0)
    lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
    span = hi - lo
    qpos = (lo - 0.05 * span) + 1.1 * span * rng.random((nenv, cm.nq))   # some beyond the limits
    qvel = rng.standard_normal((nenv, cm.nv)) * 2.0
    act = rng.random((nenv, cm.na))
    ctrl = rng.random((nenv, cm.nu))
    st = E.BatchState(hm, nenv)
    st.qpos.copy_(torch.from_numpy(qpos.astype(np.float32)))
    st.qvel.copy_(torch.from_numpy(qvel.astype(np.float32)))
    st.act.copy_(torch.from_numpy(act.astype(np.float32)))
    tctrl = torch.from_numpy(ctrl.astype(np.float32)).cuda()
    dump = E.debug_dump(hm, st, tctrl).cpu().numpy()
    names = ["xpos", "xquat", "xipos", "cdof", "cvel", "tenlen", "tenvel", "actfrc", "actdot", "bias", "smooth",
             "qaccsm", "qacc", "qfrccon"]
    omap = {"tenlen": "ten_length", "tenvel": "ten_velocity", "actfrc": "actuator_force", "actdot": "act_dot",
            "bias": "qfrc_bias", "smooth": "qfrc_smooth", "qaccsm": "qacc_smooth", "qfrccon": "qfrc_constraint"}
    worst = {n: (0.0, 0.0) for n in names}
    worst["tenJ"] = (0.0, 0.0); worst["M"] = (0.0, 0.0)
    for e in range(nenv):
        d = O.OracleData(om)
        d.qpos[:] = qpos[e].astype(np.float32); d.qvel[:] = qvel[e].astype(np.float32)
        d.act[:] = act[e].astype(np.float32); d.ctrl[:] = ctrl[e].astype(np.float32)
        d.forward()
        for n in names:
            ref = getattr(d, omap.get(n, n)).ravel()
            off = hm.layout(n)
            got = dump[e, off:off + ref.size]
            a, r = rel(got, ref)
            if r > worst[n][1]:
                worst[n] = (a, r)
        M = dump[e, hm.layout("M"):hm.layout("M") + cm.nv * cm.nv].reshape(cm.nv, cm.nv)
        a, r = rel(M, d.full_M())
        if r > worst["M"][1]:
            worst["M"] = (a, r)
        tj = dump[e, hm.layout("tenj"):hm.layout("tenj") + cm.ntenJ]
        adr = cm.arrays["TENJ_ADR"]; dof = cm.arrays["TENJ_DOF"]
        dense = np.zeros((cm.ntendon, cm.nv))
        for t in range(cm.ntendon):
            for k in range(adr[t], adr[t + 1]):
                dense[t, dof[k]] = tj[k]
        a, r = rel(dense, d.ten_J)
        if r > worst["tenJ"][1]:
            worst["tenJ"] = (a, r)
    for n, (a, r) in worst.items():
        flag = "" if r < 2e-4 else "   <<<<<<"
        print(f"  {n:10s} max abs {a:.3e}  rel {r:.3e}{flag}")
    # ---- rollout: 30 env-steps x 10 substeps with the same ctrl stream
    nsteps, nsub = 30, 10
    st = E.BatchState(hm, nenv)
    q0 = lo + span * rng.random((nenv, cm.nq))
    st.qpos.copy_(torch.from_numpy(q0.astype(np.float32)))
    ds = []
    for e in range(nenv):
        d = O.OracleData(om); d.qpos[:] = q0[e].astype(np.float32); ds.append(d)
    actions = rng.random((nsteps, nenv, cm.nu)).astype(np.float32)
    ctrls = 1.0 / (1.0 + np.exp(-5.0 * (actions.astype(np.float64) - 0.5)))
    errs = []
    for s in range(nsteps):
        E.step(hm, st, torch.from_numpy(ctrls[s].astype(np.float32)).cuda(), nsub)
        for e in range(nenv):
            ds[e].ctrl[:] = ctrls[s, e].astype(np.float32)
            ds[e].step(nsub)
        gq = st.qpos.cpu().numpy()
        oq = np.stack([d.qpos for d in ds])
        errs.append(np.abs(gq - oq).max())
    print("  rollout max|dqpos| per env-step:", " ".join(f"{x:.1e}" for x in errs[::3]))
    print("  status:", st.status.cpu().numpy().max(), " oracle warn:", max(d.warn for d in ds))
    # ---- quick timing
    nenv = 4096
    st = E.BatchState(hm, nenv)
    a = torch.rand(nenv, cm.nu, device="cuda")
    for _ in range(3):
        E.step(hm, st, a, 10)
    torch.cuda.synchronize()
    t = time.time()
    n = 20
    for _ in range(n):
        E.step(hm, st, a, 10)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"  timing: {nenv} envs x 10 substeps: {dt*1e3:.3f} ms -> {nenv/dt/1e6:.2f} M env-steps/s")


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "elbow"
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    main(name, lanes)
