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
ds = []
for e in range(nenv):
    d = O.OracleData(om); d.qpos[:] = q0[e]; d.ctrl[:] = ctrl[e]; ds.append(d)
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    for d in ds: d.step(1)
# now copy oracle state (incl. warmstart) to GPU and do ONE forward on both
st = E.BatchState(hm, nenv)
st.qpos.copy_(torch.from_numpy(np.stack([d.qpos for d in ds]).astype(np.float32)))
st.qvel.copy_(torch.from_numpy(np.stack([d.qvel for d in ds]).astype(np.float32)))
st.act.copy_(torch.from_numpy(np.stack([d.act for d in ds]).astype(np.float32)))
st.qacc_warmstart.copy_(torch.from_numpy(np.stack([d.qacc_warmstart for d in ds]).astype(np.float32)))
der = E.Derived(hm, nenv, ["qacc", "nefc", "solver_niter"])
dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda())
E.forward(hm, st, torch.from_numpy(ctrl).cuda(), der)
torch.cuda.synchronize()
dump = dump.cpu().numpy()
for e, d in enumerate(ds):
    # oracle forward from the same (float32-rounded) state
    d.qpos[:] = d.qpos.astype(np.float32); d.qvel[:] = d.qvel.astype(np.float32); d.act[:] = d.act.astype(np.float32)
    d.qacc_warmstart[:] = d.qacc_warmstart.astype(np.float32)
    d.forward()
    gq = der["qacc"][e].cpu().numpy()
    err = np.abs(gq - d.qacc).max()
    if err > 1e-2 * max(1, np.abs(d.qacc).max()):
        print("env", e, "nefc gpu/ora", int(der["nefc"][e]), d.nefc, "niter gpu/ora", int(der["solver_niter"][e]), d.solver_niter, "err", err)
        print(" gpu qacc", gq); print(" ora qacc", np.array(d.qacc)); print(" ora qaccsm", np.array(d.qacc_smooth))
        o = hm.layout("qaccsm"); print(" gpu qaccsm", dump[e, o:o+cm.nv])
        o = hm.layout("efc_active"); act = dump[e, o:o+64]; print(" gpu active lanes", np.nonzero(act)[0])
        o = hm.layout("efc_D"); print(" gpu D", dump[e, o:o+64][act > 0]); o = hm.layout("efc_aref"); print(" gpu aref", dump[e, o:o+64][act > 0])
        print(" ora D", d.efc_D[:d.nefc], "aref", d.efc_aref[:d.nefc], "frc", d.efc_force[:d.nefc])
print("max nefc", max(d.nefc for d in ds), "done")
