import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
cm = synth.get_model("hand"); om = O.OracleModel(cm)
g = np.load("tests/golden/oracle_traj_hand.npz")
nenv = g["qpos"].shape[1]
out = {}
for (s, env) in ((9, 7), (22, 0)):
    a = EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu)
    ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
    tc = torch.from_numpy(ctrl[env:env+1]).cuda()
    hm32 = E.HipModel(cm, lanes_per_env=32); hm64 = E.HipModel(cm, lanes_per_env=64)
    # find the first substep where G=32 and G=64 disagree
    def run(hm, n):
        st = E.BatchState(hm, 1)
        st.qpos.copy_(torch.from_numpy(g["qpos"][s, env:env+1].astype(np.float32))); st.qvel.copy_(torch.from_numpy(g["qvel"][s, env:env+1].astype(np.float32))); st.act.copy_(torch.from_numpy(g["act"][s, env:env+1].astype(np.float32)))
        if n: E.step(hm, st, tc, n)
        return st
    for n in range(1, 11):
        v32 = run(hm32, n).qvel.cpu().numpy(); v64 = run(hm64, n).qvel.cpu().numpy()
        if np.abs(v32 - v64).max() > 1e-3:
            break
    print("case", s, env, "first bad substep", n, "diff", np.abs(v32 - v64).max())
    st = run(hm32, n - 1)
    pre = [x.clone() for x in (st.qpos, st.qvel, st.act, st.qacc_warmstart)]
    for hm, nm in ((hm32, "g32"), (hm64, "g64")):
        s2 = E.BatchState(hm, 1)
        for dst, src in zip((s2.qpos, s2.qvel, s2.act, s2.qacc_warmstart), pre): dst.copy_(src)
        dump = E.debug_dump(hm, s2, tc).cpu().numpy()[0]
        for f in ("M", "smooth", "qaccsm", "qacc", "qfrccon", "efc_active", "efc_D", "efc_aref"):
            n_ = {"M": cm.nv * cm.nv, "efc_active": 64, "efc_D": 64, "efc_aref": 64}.get(f, cm.nv)
            out[f"{s}_{env}_{nm}_{f}"] = dump[hm.layout(f):hm.layout(f) + n_]
        out[f"{s}_{env}_{nm}_niter"] = dump[hm.layout("scal")]
    out[f"{s}_{env}_warm"] = pre[3].cpu().numpy()[0]; out[f"{s}_{env}_qpos"] = pre[0].cpu().numpy()[0]; out[f"{s}_{env}_qvel"] = pre[1].cpu().numpy()[0]
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/newton_cases.npz", **out)
