"""Where does the kernel's tendon Jacobian lose accuracy?  Worst entries of ten_J (hand, 64 random states) against the fp64
oracle, with what kind of entry each is (tendon, dof, joint type, wrapped or not at that state, |J| of the entry, the largest |J|
of the row).   python tests/tools/gpu_tenj_diag.py  (GPU box)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from oracle import oracle as O

cm = synth.get_model("hand"); hm = E.HipModel(cm, lanes_per_env=32); om = O.OracleModel(cm)
nenv = 64
rng = np.random.default_rng(0)
lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
qpos = (lo + (hi - lo) * rng.random((nenv, cm.nq))).astype(np.float32)
qvel = (rng.standard_normal((nenv, cm.nv)) * 2).astype(np.float32)
act = rng.random((nenv, cm.na)).astype(np.float32); ctrl = rng.random((nenv, cm.nu)).astype(np.float32)
st = E.BatchState(hm, nenv)
st.qpos.copy_(torch.from_numpy(qpos)); st.qvel.copy_(torch.from_numpy(qvel)); st.act.copy_(torch.from_numpy(act))
dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda()).cpu().numpy()
adr = cm.arrays["TENJ_ADR"]; dof = cm.arrays["TENJ_DOF"]
tadr, tnum = cm.arrays["TENDON_ADR"], cm.arrays["TENDON_NUM"]; wtype = cm.arrays["WRAP_TYPE"]
rows = []
jmax = 0.0
lenerr = []
for e in range(nenv):
    d = O.OracleData(om); d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]; d.forward()
    tj = dump[e, hm.layout("tenj"):hm.layout("tenj") + len(dof)]
    tl = dump[e, hm.layout("tenlen"):hm.layout("tenlen") + cm.ntendon]
    J = d.ten_J
    jmax = max(jmax, np.abs(J).max())
    for t in range(cm.ntendon):
        lenerr.append((abs(tl[t] - d.ten_length[t]), t, e))
        nwrapgeom = int((wtype[tadr[t]:tadr[t] + tnum[t]] >= 4).sum())
        for k in range(adr[t], adr[t + 1]):
            ref = J[t, dof[k]]
            rows.append((abs(tj[k] - ref), e, t, int(dof[k]), float(ref), float(tj[k]), float(np.abs(J[t]).max()), nwrapgeom))
rows.sort(reverse=True)
print("max |J| over everything", jmax, " wrap-type codes present", sorted(set(wtype.tolist())))
print("worst 25 entries: abs err, env, tendon, dof, oracle, gpu, row max |J|, wrap geoms on the tendon")
for r in rows[:25]:
    print(f"{r[0]:.2e} env {r[1]:2d} tendon {r[2]:2d} dof {r[3]:2d} ref {r[4]:+.6f} gpu {r[5]:+.6f} rowmax {r[6]:.4f} wraps {r[7]}")
err = np.array([r[0] for r in rows]); wr = np.array([r[7] for r in rows]) > 0
print("entries on tendons WITH wrap geoms: n", int(wr.sum()), "max err", err[wr].max() if wr.any() else 0, "rms", np.sqrt((err[wr] ** 2).mean()) if wr.any() else 0)
print("entries on tendons WITHOUT wrap geoms: n", int((~wr).sum()), "max err", err[~wr].max(), "rms", np.sqrt((err[~wr] ** 2).mean()))
bydof = {}
for r in rows:
    bydof.setdefault(r[3], []).append(r[0])
print("max err by dof:", {k: f"{max(v):.1e}" for k, v in sorted(bydof.items())})
lenerr.sort(reverse=True)
print("worst tendon length errors:", [(f"{a:.2e}", t, e) for a, t, e in lenerr[:8]])
