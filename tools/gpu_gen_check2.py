import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from oracle import oracle as O
from tools.gpu_gen_check import states
name = "leg"
cm = synth.get_model(name); hm = E.HipModel(cm); om = O.OracleModel(cm)
rng = np.random.default_rng(1)
n = 24
q, v = states(cm, name, n, rng)
act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.random((n, cm.nu)).astype(np.float32)
st = E.BatchState(hm, n)
st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v)); st.act.copy_(torch.from_numpy(act))
ds = []
for e in range(n):
    d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]; ds.append(d)
c = torch.from_numpy(ctrl).cuda()
jn = list(cm.names["joint"].keys())
for sub in range(30):
    E.step(hm, st, c, 1)
    for d in ds: d.step(1)
    qg = st.qpos.cpu().numpy(); qo = np.array([d.qpos for d in ds])
    err = np.abs(qg - qo)
    e = int(err.max(axis=1).argmax()); i = int(err[e].argmax())
    print(sub, f"max err {err.max():.2e} env {e} qpos[{i}]", "oracle nefc", ds[e].nefc, "niter", ds[e].solver_niter, "| median", f"{np.median(err.max(axis=1)):.2e}")
