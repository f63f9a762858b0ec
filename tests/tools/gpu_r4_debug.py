#!/usr/bin/env python
"""round-4 debugging on the GPU box: plane_toy stage parity, warm-start sensitivity of the contact solves"""
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

name = "plane_toy"
cm = synth.get_model(name); hm = E.HipModel(cm); om = O.OracleModel(cm)
rng = np.random.default_rng(1)
n = 24
q, v = T._states(cm, name, n, rng)
ctrl = rng.random((n, cm.nu)).astype(np.float32)
st = E.BatchState(hm, n)
st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
d_ = E.Derived(hm, n, ["qacc", "nefc", "solver_niter"])
dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda()).cpu().numpy()
E.forward(hm, st, torch.from_numpy(ctrl).cuda(), d_)
for e in range(n):
    d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.ctrl[:] = ctrl[e]; d.forward()
    got = dump[e, hm.layout("qacc"):hm.layout("qacc") + cm.nv]
    gsm = dump[e, hm.layout("qaccsm"):hm.layout("qaccsm") + cm.nv]
    pp = np.zeros(cm.npair, int)
    for c in d.con_pair: pp[c] += 1
    print(f"env {e}: oracle nefc {d.nefc} niter {d.solver_niter} contacts/pair {pp.tolist()} | gpu nefc {int(d_['nefc'][e])} niter {int(d_['solver_niter'][e])} "
          f"| rel err qacc_smooth {np.abs(gsm - d.qacc_smooth).max() / max(1e-9, np.abs(d.qacc_smooth).max()):.1e} qacc {np.abs(got - d.qacc).max() / max(1e-9, np.abs(d.qacc).max()):.1e} status {int(st.status[e])}")

print("---- warm-start sensitivity")
for env_id, kw in (("myoHandPoseRandom-v0", {"model": "hand_contact"}), ("myoHandReorient100-v0", {})):
    env = registry.make(env_id, num_envs=128, seed=5, autoreset=True, **kw)
    env.rollout_setup(action_seed=9)
    for s in range(9):
        env.rollout_step(None, stream_id=s)
    st, hm = env.state, env.hm
    n = 128
    f = ["qacc", "nefc", "solver_niter"]
    d1, d2 = E.Derived(hm, n, f), E.Derived(hm, n, f)
    ctrl = env.last_ctrl.clone(); keep = st.qacc_warmstart.clone()
    stat0 = st.status.clone()
    E.forward(hm, st, ctrl, d1); s1 = st.status.clone()
    st.qacc_warmstart.zero_(); st.status.copy_(stat0)
    E.forward(hm, st, ctrl, d2); s2 = st.status.clone()
    st.qacc_warmstart.copy_(keep)
    qa, qb = d1["qacc"].double(), d2["qacc"].double()
    rel = ((qa - qb).abs().amax(dim=1) / qa.abs().amax(dim=1).clamp(min=1.0)).cpu().numpy()
    order = np.argsort(-rel)[:8]
    om = O.OracleModel(env.cm)
    for e in order:
        d = O.OracleData(om)
        if "Reorient" in env_id:
            d.set_geom_size(env.cm.names["geom"]["obj"], env.geom_size[e].cpu().numpy().astype(np.float64), int(env.geom_type[e]))
        d.qpos[:] = st.qpos[e].cpu().numpy(); d.qvel[:] = st.qvel[e].cpu().numpy(); d.act[:] = st.act[e].cpu().numpy(); d.ctrl[:] = ctrl[e].cpu().numpy()
        d.qacc_warmstart[:] = keep[e].cpu().numpy(); d.forward()
        qo = d.qacc.copy(); n1 = d.solver_niter
        d.qacc_warmstart[:] = 0; d.forward()
        qo0 = d.qacc.copy()
        sc = max(1.0, np.abs(qo).max())
        print(f"{env_id} env {e}: gpu rel diff {rel[e]:.2e}, nefc {int(d1['nefc'][e])}, niter stepped/zero {int(d1['solver_niter'][e])}/{int(d2['solver_niter'][e])}, status {int(s1[e])}/{int(s2[e])}; "
              f"oracle niter {n1}/{d.solver_niter} oracle stepped-vs-zero {np.abs(qo - qo0).max() / sc:.2e}; gpu-vs-oracle stepped {np.abs(qa[e].cpu().numpy() - qo).max() / sc:.2e} zero {np.abs(qb[e].cpu().numpy() - qo0).max() / sc:.2e}")
