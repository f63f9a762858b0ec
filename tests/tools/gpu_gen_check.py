"""GEN (contact/equality) path check on the GPU: forward + rollouts of contact_toy and leg vs the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.model import synth
from oracle import oracle as O

def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(1e-9, np.abs(b).max()))

def states(cm, name, n, rng):
    q = np.tile(cm.qpos0.astype(np.float64), (n, 1)); v = np.zeros((n, cm.nv))
    if name == "contact_toy":
        q[:, 2] += rng.uniform(-0.04, 0.05, n)            # log height: in and out of contact
        qq = rng.standard_normal((n, 4)) * 0.2 + np.array([1, 0, 0, 0]); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        q[:, 7:10] += rng.uniform(-0.05, 0.05, (n, 3)); q[:, 9] -= rng.uniform(0.0, 0.2, n)
        q[:, 10] = rng.uniform(-1.3, 1.3, n); q[:, 11] = rng.uniform(-0.45, 0.12, n); q[:, 12] = rng.uniform(-0.1, 0.1, n)
        v = rng.standard_normal((n, cm.nv)) * 0.5
    else:
        k = cm.key_qpos
        for e in range(n):
            q[e] = k[(0, 2, 3)[e % 3]]
        q[:, 7:] += rng.uniform(-0.15, 0.15, (n, cm.nq - 7))
        q[:, 2] += rng.uniform(-0.03, 0.02, n)
        qq = rng.standard_normal((n, 4)) * 0.05 + np.array([1, 0, 0, 0]); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        v = rng.standard_normal((n, cm.nv)) * 0.3
    return q.astype(np.float32), v.astype(np.float32)

def main():
  for name in ("contact_toy", "leg"):
      cm = synth.get_model(name); hm = E.HipModel(cm); om = O.OracleModel(cm)
      print(name, "lanes", hm.info(E.INFO_LANES), "lds/env", hm.info(E.INFO_LDS_PER_ENV))
      rng = np.random.default_rng(1)
      n = 24
      q, v = states(cm, name, n, rng)
      act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.random((n, cm.nu)).astype(np.float32)
      st = E.BatchState(hm, n)
      st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v)); st.act.copy_(torch.from_numpy(act))
      dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda()).cpu().numpy()
      worst = {}
      for e in range(n):
          d = O.OracleData(om)
          d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]
          d.forward()
          for nm, ref in (("qaccsm", d.qacc_smooth), ("qacc", d.qacc), ("smooth", d.qfrc_smooth), ("bias", d.qfrc_bias)):
              got = dump[e, hm.layout(nm):hm.layout(nm) + ref.size]
              worst[nm] = max(worst.get(nm, 0), rel(got, ref))
          got = dump[e, hm.layout("qfrccon"):hm.layout("qfrccon") + cm.nv]
          worst["qfrccon_abs/scale"] = max(worst.get("qfrccon_abs/scale", 0), np.abs(got - d.qfrc_constraint).max() / max(1.0, np.abs(d.qfrc_smooth).max()))
          nit = dump[e, hm.layout("scal")]
          if e < 6: print("  env", e, "oracle nefc", d.nefc, "ncon", d.ncon, "niter", d.solver_niter, "| gpu niter", nit,
                          "qacc rel", rel(dump[e, hm.layout("qacc"):hm.layout("qacc") + cm.nv], d.qacc))
      print("  forward worst:", {k: float(f"{x:.2e}") for k, x in worst.items()})
      # free-running rollout, constant ctrl
      st = E.BatchState(hm, n)
      st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v)); st.act.copy_(torch.from_numpy(act))
      ds = []
      for e in range(n):
          d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]; ds.append(d)
      c = torch.from_numpy(ctrl).cuda()
      for blk in range(6):
          E.step(hm, st, c, 25)
          for d in ds: d.step(25)
          qg = st.qpos.cpu().numpy(); qo = np.array([d.qpos for d in ds])
          err = np.abs(qg - qo).max(axis=1)
          print(f"  after {25*(blk+1)} substeps: qpos abs err median {np.median(err):.2e} max {err.max():.2e}; status", st.status.cpu().numpy().max(), "oracle warn", max(d.warn for d in ds))

if __name__ == "__main__":
    main()
