import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
cm = synth.get_model("hand")
g = np.load("tests/golden/oracle_traj_hand.npz")
nenv = g["qpos"].shape[1]; s = 9
a = EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu)
ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
tc = torch.from_numpy(ctrl[7:8]).cuda()
hm = E.HipModel(cm, lanes_per_env=int(sys.argv[1]))
st = E.BatchState(hm, 1)
st.qpos.copy_(torch.from_numpy(g["qpos"][s, 7:8].astype(np.float32))); st.qvel.copy_(torch.from_numpy(g["qvel"][s, 7:8].astype(np.float32))); st.act.copy_(torch.from_numpy(g["act"][s, 7:8].astype(np.float32)))
E.step(hm, st, tc, 9)
torch.cuda.synchronize(); print("=== substep 10", flush=True)
E.step(hm, st, tc, 1)
torch.cuda.synchronize()
print("qvel[16:]", st.qvel.cpu().numpy()[0, 16:])
