"""Free-running divergence vs the fp64 oracle over 100 env-steps = 1000 physics steps (north_star accuracy metric).
python tools/parity_1000.py [nenv]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O, env_oracle as EO
nenv = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out = {}
for name in ("elbow", "hand"):
    cm = synth.get_model(name); om = O.OracleModel(cm); hm = E.HipModel(cm)
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    q0 = np.stack([(lo + (hi - lo) * EO.pose_reset_draws(cm.nq, e, 0, 0)[0]).astype(np.float32) for e in range(nenv)])
    st = E.BatchState(hm, nenv); st.qpos.copy_(torch.from_numpy(q0))
    ds = []
    for e in range(nenv):
        d = O.OracleData(om); d.qpos[:] = q0[e]; ds.append(d)
    a = torch.empty(nenv, cm.nu, device="cuda")
    rel = []
    for s in range(100):
        E.uniform(a, 0, s)
        ctrl = (1.0 / (1.0 + torch.exp(-5.0 * (a - 0.5)))).contiguous()
        E.step(hm, st, ctrl, 10)
        c = ctrl.cpu().numpy()
        for e, d in enumerate(ds):
            d.ctrl[:] = c[e]; d.step(10)
        oq = np.stack([d.qpos for d in ds]); gq = st.qpos.cpu().numpy()
        rel.append(float(np.abs(gq - oq).max() / max(1.0, np.abs(oq).max())))
    per_env = np.abs(st.qpos.cpu().numpy() - np.stack([d.qpos for d in ds])).max(axis=1)
    out[name] = dict(rel_err_at_100_300_1000_steps=[rel[9], rel[29], rel[99]], max_over_run=max(rel),
                     envs_below_1e4=int((per_env < 1e-4 * max(1.0, np.abs(oq).max())).sum()), nenv=nenv,
                     status=int(st.status.max()))
    print(name, json.dumps(out[name]))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/parity_1000.json", "w"), indent=1)
