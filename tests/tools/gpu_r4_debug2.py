#!/usr/bin/env python
"""round-4 debugging: is a GPU contact solve that differs from the oracle a different PROBLEM (rows) or an unconverged SOLVE?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
from myosuite_amd.model import synth
from oracle import oracle as O
import test_contacts as T


def analyse(tag, cm, hm, om, q, v, act, ctrl, warm, setup=None):
    n = 1
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q[None].astype(np.float32))); st.qvel.copy_(torch.from_numpy(v[None].astype(np.float32)))
    if cm.na:
        st.act.copy_(torch.from_numpy(act[None].astype(np.float32)))
    st.qacc_warmstart.copy_(torch.from_numpy(warm[None].astype(np.float32)))
    if setup:
        setup(st)
    c = torch.from_numpy(ctrl[None].astype(np.float32)).cuda()
    dump = E.debug_dump(hm, st, c).cpu().numpy()[0]
    L = hm.layout
    d = O.OracleData(om)
    if setup:
        setup(d)
    d.qpos[:] = q; d.qvel[:] = v
    if cm.na:
        d.act[:] = act
    d.ctrl[:] = ctrl; d.qacc_warmstart[:] = warm
    d.forward()
    nefc, nv = d.nefc, cm.nv
    gD = dump[L("efc_D"):L("efc_D") + 64][:nefc]; gA = dump[L("efc_aref"):L("efc_aref") + 64][:nefc]
    relD = np.abs(gD - d.efc_D[:nefc]) / np.abs(d.efc_D[:nefc]); relA = np.abs(gA - d.efc_aref[:nefc]) / np.maximum(1.0, np.abs(d.efc_aref[:nefc]))
    print(f"{tag}: nefc {nefc}, oracle niter {d.solver_niter}, gpu niter {int(dump[L('scal')])}; row D rel err max {relD.max():.1e}, aref rel err max {relA.max():.1e} (row {int(np.argmax(relA))})")
    J = d.efc_J[:nefc].copy(); D = d.efc_D[:nefc].copy(); aref = d.efc_aref[:nefc].copy()
    M = np.zeros((nv, nv)); O.lib().mmo_full_m(om.ptr, d.ptr, M.ctypes.data)
    fs = d.qfrc_smooth.copy(); a0 = d.qacc_smooth.copy()

    def cost_grad(a):
        jar = J @ a - aref
        on = jar < 0
        f = np.where(on, -D * jar, 0.0)
        grad = M @ (a - a0) - J.T @ f
        cost = 0.5 * (a - a0) @ (M @ (a - a0)) + 0.5 * np.sum(D * np.minimum(jar, 0) ** 2)
        return cost, grad, on
    ga = dump[L("qacc"):L("qacc") + nv].astype(np.float64)
    co, gro, ono = cost_grad(d.qacc)
    cg, grg, ong = cost_grad(ga)
    print(f"   oracle: cost {co:.6e} |grad| {np.linalg.norm(gro):.2e};  gpu qacc in the ORACLE's problem: cost {cg:.6e} (+{(cg - co) / abs(co):.2e} rel) |grad| {np.linalg.norm(grg):.2e} "
          f"vs |M(a-a0)| {np.linalg.norm(M @ (ga - a0)):.2e}; active set differs in {int((ono != ong).sum())} rows; max|dqacc| {np.abs(ga - d.qacc).max():.2e} of {np.abs(d.qacc).max():.2e} at dof {int(np.argmax(np.abs(ga - d.qacc)))} (M_ii {M[int(np.argmax(np.abs(ga - d.qacc)))][int(np.argmax(np.abs(ga - d.qacc)))]:.2e})")
    if os.environ.get("MYOSIM_LIB"):      # trace build: the solver's running jar per row against J a - aref of its own final qacc
        gj = dump[L("efc_active"):L("efc_active") + 64][:nefc].astype(np.float64)
        jar_true = J @ ga - aref
        bad = np.argsort(-np.abs(gj - jar_true))[:6]
        print("   tracked jar vs J qacc - aref (oracle J): worst rows", [(int(r), float(f"{gj[r]:.4g}"), float(f"{jar_true[r]:.4g}")) for r in bad], " row types", d.efc_type[bad].tolist())
        rec = dump[L("scal") + 80: L("scal") + 96].astype(np.float64)
        print("   rows 16..31: tracked / recomputed on the GPU / true:", [(r, float(f"{gj[r]:.4g}"), float(f"{rec[r - 16]:.4g}"), float(f"{jar_true[r]:.4g}")) for r in range(16, min(nefc, 32))])
        r0 = max(0, nefc - 8)
        gJ = dump[L("M"):L("M") + 8 * nv].reshape(8, nv).astype(np.float64)
        for r in range(min(8, nefc)):
            dj = np.abs(gJ[r] - J[r0 + r]).max()
            if dj > 1e-4 * max(1.0, np.abs(J[r0 + r]).max()):
                print(f"   J row {r0 + r}: max|gpu - oracle| {dj:.3e}; gpu {np.round(gJ[r], 4).tolist()}")
                print(f"                                      oracle {np.round(J[r0 + r], 4).tolist()}")
        gq = dump[L("qfrccon"):L("qfrccon") + nv].astype(np.float64)
        f_true = np.where(jar_true < 0, -D * jar_true, 0.0)
        print(f"   gpu qfrc_constraint vs J'f(oracle rows at gpu qacc): max diff {np.abs(gq - J.T @ f_true).max():.3e} of {np.abs(gq).max():.3e}; gpu's own residual |M a - smooth - qfrccon| {np.linalg.norm(M @ ga - fs - gq):.2e}")
    w = np.linalg.eigvalsh(M + J[ono].T @ (D[ono, None] * J[ono]))
    print(f"   cond(H) at the optimum {w.max() / w.min():.2e}; cond(M) {np.linalg.cond(M):.2e}")


# plane_toy envs 0, 8
cm = synth.get_model("plane_toy"); hm = E.HipModel(cm); om = O.OracleModel(cm)
rng = np.random.default_rng(1)
q, v = T._states(cm, "plane_toy", 24, rng)
ctrl = rng.random((24, cm.nu)).astype(np.float32)
for e in (0, 8, 9, 19):
    analyse(f"plane_toy env {e}", cm, hm, om, q[e].astype(np.float64), v[e].astype(np.float64), np.zeros(0), ctrl[e].astype(np.float64), np.zeros(cm.nv))

env = registry.make("myoHandPoseRandom-v0", num_envs=128, seed=5, autoreset=True, model="hand_contact")
env.rollout_setup(action_seed=9)
for s in range(9):
    env.rollout_step(None, stream_id=s)
st = env.state
hm2 = E.HipModel(env.cm, lanes_per_env=64); om2 = O.OracleModel(env.cm)
for e in (110, 48):
    for wtag, w in (("stepped", st.qacc_warmstart[e].cpu().numpy().astype(np.float64)), ("zero", np.zeros(env.cm.nv))):
        analyse(f"hand_contact env {e} warm={wtag}", env.cm, hm2, om2, st.qpos[e].cpu().numpy().astype(np.float64), st.qvel[e].cpu().numpy().astype(np.float64),
                st.act[e].cpu().numpy().astype(np.float64), env.last_ctrl[e].cpu().numpy().astype(np.float64), w)

# per-iteration trace (MM_NEWTON_TRACE build): gn, done, alpha, q1, |search|, active rows
for e in (110, 48):
    for wtag, w in (("stepped", st.qacc_warmstart[e].cpu().numpy()), ("zero", np.zeros(env.cm.nv, np.float32))):
        s1 = E.BatchState(hm2, 1)
        s1.qpos.copy_(st.qpos[e:e + 1]); s1.qvel.copy_(st.qvel[e:e + 1]); s1.act.copy_(st.act[e:e + 1]); s1.qacc_warmstart.copy_(torch.from_numpy(w[None].astype(np.float32)))
        dump = E.debug_dump(hm2, s1, env.last_ctrl[e:e + 1].clone()).cpu().numpy()[0]
        tr = dump[hm2.layout("scal") + 32: hm2.layout("scal") + 32 + 48].reshape(8, 6)
        print(f"trace env {e} warm={wtag}: niter {int(dump[hm2.layout('scal')])}")
        for it in range(8):
            print(f"   iter {it}: |grad| {tr[it, 0]:.4e} done {tr[it, 1]:.0f} alpha {tr[it, 2]:.6f} q1 {tr[it, 3]:.4e} |search| {tr[it, 4]:.4e} active {tr[it, 5]:.0f}")
