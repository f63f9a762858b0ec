import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
cm = synth.get_model("hand"); om = O.OracleModel(cm)
g = np.load("tests/golden/oracle_traj_hand.npz")
nenv = g["qpos"].shape[1]; s = 9
a = EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu)
ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
hm = E.HipModel(cm, lanes_per_env=32)
def run(idx, nsub):
    n = len(idx)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(g["qpos"][s][idx].astype(np.float32))); st.qvel.copy_(torch.from_numpy(g["qvel"][s][idx].astype(np.float32)))
    st.act.copy_(torch.from_numpy(g["act"][s][idx].astype(np.float32)))
    E.step(hm, st, torch.from_numpy(ctrl[idx]).cuda(), nsub)
    return st.qvel.cpu().numpy(), st.status.cpu().numpy()
d = O.OracleData(om)
for nsub in range(1, 11):
    d.reset(); d.qpos[:] = g["qpos"][s, 7].astype(np.float32); d.qvel[:] = g["qvel"][s, 7].astype(np.float32); d.act[:] = g["act"][s, 7].astype(np.float32); d.ctrl[:] = ctrl[7]
    d.step(nsub)
    v_pair, st1 = run([6, 7], nsub); v_same, st2 = run([7, 7], nsub); v_single, st3 = run([7], nsub); v_swapped, _ = run([7, 6], nsub)
    print(nsub, "nefc", d.nefc, "niter", d.solver_niter, "dv pair(6,7)", np.abs(v_pair[1]-d.qvel).max(), "same(7,7)", np.abs(v_same[1]-d.qvel).max(), "single", np.abs(v_single[0]-d.qvel).max(), "swapped(7,6)", np.abs(v_swapped[0]-d.qvel).max(), st1, st2)
