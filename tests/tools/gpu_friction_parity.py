"""Debug: teacher-forced friction_toy steps, report the worst qacc mismatches (GPU forward vs oracle forward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from oracle import oracle as O
cm = synth.get_model("friction_toy")
hm = E.HipModel(cm); om = O.OracleModel(cm)
n = 16
rng = np.random.default_rng(4)
q = rng.uniform(-0.5, 0.8, (n, cm.nq)); q[:, 3] = 0.05 * q[:, 1]
v = rng.standard_normal((n, cm.nv)) * np.array([1.0, 1.0, 3.0, 0.05]); v[: n // 4] = 0.0
st = E.BatchState(hm, n)
ds = [O.OracleData(om) for _ in range(n)]
dv = E.Derived(hm, n, ["qacc", "nefc", "solver_niter"])
rows = []
for k in range(60):
    ctrl = rng.uniform(-1, 1, (n, cm.nu)).astype(np.float32)
    if k % 3 == 0: ctrl *= 0.05
    st.qpos.copy_(torch.from_numpy(q.astype(np.float32))); st.qvel.copy_(torch.from_numpy(v.astype(np.float32)))
    c = torch.from_numpy(ctrl).cuda()
    ws = st.qacc_warmstart.clone()
    E.forward(hm, st, c, dv)
    qa = dv["qacc"].cpu().numpy().copy(); ni = dv["solver_niter"].cpu().numpy().copy()
    E.step(hm, st, c, 1)
    for e, d in enumerate(ds):
        d.qpos[:] = q[e].astype(np.float32); d.qvel[:] = v[e].astype(np.float32); d.ctrl[:] = ctrl[e]
        d.qacc_warmstart[:] = ws[e].cpu().numpy()
        d.forward()
        err = np.abs(qa[e] - d.qacc)
        rows.append((err.max(), k, e, int(err.argmax()), qa[e].copy(), d.qacc.copy(), int(ni[e]), d.solver_niter, d.efc_force[:d.nefc].copy(), int(st.status[e])))
        d.step(1)
        q[e] = d.qpos; v[e] = d.qvel
rows.sort(key=lambda r: -r[0])
for r in rows[:6]:
    print("err %.4f step %d env %d dof %d niter gpu %d cpu %d status %d" % (r[0], r[1], r[2], r[3], r[6], r[7], r[9]))
    print("   gpu", r[4], "\n   cpu", np.round(r[5], 4), "\n   force", np.round(r[8], 5))
