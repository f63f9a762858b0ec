import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O
np.set_printoptions(precision=4, suppress=True, linewidth=200)
cm = synth.get_model("hand"); om = O.OracleModel(cm); hm = E.HipModel(cm)
rng = np.random.default_rng(0)
lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
nenv = 37
q0 = (lo + (hi - lo) * rng.random((nenv, cm.nq))).astype(np.float32)
ctrl = rng.random((nenv, cm.nu)).astype(np.float32)
st = E.BatchState(hm, nenv); st.qpos.copy_(torch.from_numpy(q0))
ds = []
for e in range(nenv):
    d = O.OracleData(om); d.qpos[:] = q0[e]; d.ctrl[:] = ctrl[e]; ds.append(d)
tc = torch.from_numpy(ctrl).cuda()
for s in range(40):
    E.step(hm, st, tc, 1)
    for d in ds: d.step(1)
    gq = st.qpos.cpu().numpy(); gv = st.qvel.cpu().numpy(); ga = st.act.cpu().numpy(); gw = st.qacc_warmstart.cpu().numpy()
    oq = np.stack([d.qpos for d in ds]); ov = np.stack([d.qvel for d in ds]); oa = np.stack([d.act for d in ds]); ow = np.stack([d.qacc_warmstart for d in ds])
    if s % 5 == 0 or not np.isfinite(gv).all() or np.abs(gv-ov).max() > 1e-2: print(s, "dq", np.abs(gq-oq).max(), "dv", np.abs(gv-ov).max(), "da", np.abs(ga-oa).max(), "dwarm", np.abs(gw-ow).max(), "nefc", max(d.nefc for d in ds), "niter", max(d.solver_niter for d in ds), "status", st.status.cpu().numpy().max())
    if not np.isfinite(gv).all() or np.abs(gv-ov).max() > 1e-2:
        e = int(np.argmax(np.abs(gv-ov).max(axis=1)))
        print(" oracle nefc", ds[e].nefc, "niter", ds[e].solver_niter, "efc_force", ds[e].efc_force[:ds[e].nefc], "qacc", np.array(ds[e].qacc))
        print(" env", e, "gpu qvel", gv[e]); print(" ora qvel", ov[e]); print(" gpu warm", gw[e]); print(" ora warm", ow[e])
        break
