import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
cm = synth.get_model("hand"); om = O.OracleModel(cm)
g = np.load("tests/golden/oracle_traj_hand.npz")
nsteps, nenv = g["qpos"].shape[0] - 1, g["qpos"].shape[1]
hm = E.HipModel(cm, lanes_per_env=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
for s in range(0, nsteps):
    a = EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu)
    ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
    st = E.BatchState(hm, nenv)
    st.qpos.copy_(torch.from_numpy(g["qpos"][s].astype(np.float32))); st.qvel.copy_(torch.from_numpy(g["qvel"][s].astype(np.float32))); st.act.copy_(torch.from_numpy(g["act"][s].astype(np.float32)))
    E.step(hm, st, torch.from_numpy(ctrl).cuda(), 10)
    stt = st.status.cpu().numpy()
    for e in range(nenv):
        d = O.OracleData(om)
        d.qpos[:] = g["qpos"][s, e].astype(np.float32); d.qvel[:] = g["qvel"][s, e].astype(np.float32); d.act[:] = g["act"][s, e].astype(np.float32); d.ctrl[:] = ctrl[e]
        d.step(10)
        dv = np.abs(st.qvel[e].cpu().numpy() - d.qvel).max()
        if dv > 5e-3 or stt[e] != 0:
            print(f"step {s} env {e}: dv {dv:.3e} status {stt[e]:#x}")
print("done")
