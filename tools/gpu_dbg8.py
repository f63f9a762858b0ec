import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
np.set_printoptions(precision=4, suppress=True, linewidth=220)
cm = synth.get_model("hand"); om = O.OracleModel(cm)
g = np.load("tests/golden/oracle_traj_hand.npz")
nenv = g["qpos"].shape[1]; s = 9
a = EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu)
ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
d = O.OracleData(om)
d.qpos[:] = g["qpos"][s, 7].astype(np.float32); d.qvel[:] = g["qvel"][s, 7].astype(np.float32); d.act[:] = g["act"][s, 7].astype(np.float32); d.ctrl[:] = ctrl[7]
d.step(9)
state = [np.array(x, dtype=np.float32) for x in (d.qpos, d.qvel, d.act, d.qacc_warmstart)]
d.forward()
print("oracle nefc", d.nefc, "niter", d.solver_niter, "rows: pos", d.efc_pos[:d.nefc], "frc", d.efc_force[:d.nefc])
print("ora qacc", np.array(d.qacc)); print("ora qfrccon", np.array(d.qfrc_constraint))
for lanes in (64, 32):
    hm = E.HipModel(cm, lanes_per_env=lanes)
    st = E.BatchState(hm, 1)
    for dst, src in zip((st.qpos, st.qvel, st.act, st.qacc_warmstart), state): dst.copy_(torch.from_numpy(src[None]))
    dump = E.debug_dump(hm, st, torch.from_numpy(ctrl[7:8]).cuda()).cpu().numpy()[0]
    o = hm.layout("qacc"); print("lanes", lanes, "qacc", dump[o:o+cm.nv]); o = hm.layout("qfrccon"); print("   qfrccon", dump[o:o+cm.nv])
    o = hm.layout("efc_active"); act = dump[o:o+64]; print("   active lanes", np.nonzero(act)[0], "niter", dump[hm.layout("scal")], "D", dump[hm.layout("efc_D"):hm.layout("efc_D")+64][act > 0])
