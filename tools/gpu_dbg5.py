import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O
np.set_printoptions(precision=4, suppress=True, linewidth=220)
cm = synth.get_model("hand"); om = O.OracleModel(cm); hm = E.HipModel(cm)
nenv = 37
rng = np.random.default_rng(0)
lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
span = hi - lo
qpos = (lo - 0.05 * span) + 1.1 * span * rng.random((nenv, cm.nq))
qvel = rng.standard_normal((nenv, cm.nv)) * 2.0
act = rng.random((nenv, cm.na)); ctrl = rng.random((nenv, cm.nu))
st = E.BatchState(hm, nenv)
st.qpos.copy_(torch.from_numpy(qpos.astype(np.float32))); st.qvel.copy_(torch.from_numpy(qvel.astype(np.float32))); st.act.copy_(torch.from_numpy(act.astype(np.float32)))
dump = E.debug_dump(hm, st, torch.from_numpy(ctrl.astype(np.float32)).cuda()).cpu().numpy()
bad = 0
for e in range(nenv):
    d = O.OracleData(om)
    d.qpos[:] = qpos[e].astype(np.float32); d.qvel[:] = qvel[e].astype(np.float32); d.act[:] = act[e].astype(np.float32); d.ctrl[:] = ctrl[e].astype(np.float32)
    d.forward()
    o = hm.layout("actfrc"); got = dump[e, o:o+cm.nu]
    err = np.abs(got - d.actuator_force)
    if err.max() > 1e-2 and bad < 3:
        bad += 1
        k = int(np.argmax(err)); print("env", e, "act", k, "gpu", got[k], "ora", d.actuator_force[k], "nbad", (err > 1e-2).sum())
        print(" gpu", got[:12]); print(" ora", np.array(d.actuator_force[:12]))
        o = hm.layout("qfrccon"); print(" gpu qfrccon", dump[e, o:o+cm.nv]); print(" ora qfrccon", np.array(d.qfrc_constraint))
        o = hm.layout("qacc"); print(" gpu qacc", dump[e, o:o+cm.nv]); print(" ora qacc", np.array(d.qacc)); print(" niter gpu", dump[e, hm.layout("scal")], "ora", d.solver_niter, "nefc", d.nefc)
print("done; envs with bad actfrc:", bad)
