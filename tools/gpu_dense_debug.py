"""Debug: the all-env solve scan on hand_dense; envs whose constrained acceleration is off against the oracle are dumped
(state + deltas) so that the same states can be re-run under another library build (MYOSIM_LIB=...).
  python tools/gpu_dense_debug.py scan   -> gpurun_out/dense_bad.npz
  python tools/gpu_dense_debug.py replay -> forward of the dumped states under the current library vs the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd import engine as E
from myosuite_amd.envs import registry
from myosuite_amd.model import synth
from oracle import oracle as O
O.build()
mode = sys.argv[1]
fn = "gpurun_out/dense_bad.npz" if (len(sys.argv) > 1 and sys.argv[1] == "scan") or os.path.exists("gpurun_out/dense_bad.npz") else "tests/golden/debug/dense_bad.npz"
cm = synth.get_model("hand_dense")
om = O.OracleModel(cm)

def oracle_forward(qpos, qvel, act, ctrl, warm, gs, gt, bm, gid, bid):
    d = O.OracleData(om)
    d.reset()
    d.set_geom_size(gid, gs, int(gt))
    if bm is not None: d.set_body_mass(bid, float(bm))
    d.qpos[:] = qpos; d.qvel[:] = qvel; d.act[:] = act; d.ctrl[:] = ctrl; d.qacc_warmstart[:] = warm
    d.forward()
    return d

if mode == "scan":
    n = 2048
    env = registry.make("myoHandReorient100-v0", num_envs=n, seed=23, model="hand_dense")
    env.rollout_setup(action_seed=3)
    for s in range(7): env.rollout_step(None, stream_id=s)
    st, hm = env.state, env.hm
    ctrl = env.last_ctrl.clone()
    gid, bid = int(st._c.geom_env_id), int(st._c.body_mass_env_id)
else:
    z = np.load(fn)
    n = z["qpos"].shape[0]
    hm = E.HipModel(cm, lanes_per_env=64)
    st = E.BatchState(hm, n)
    for k in ("qpos", "qvel", "act", "qacc_warmstart"): getattr(st, k).copy_(torch.from_numpy(z[k]))
    gid, bid = int(z["gid"]), int(z["bid"])
    st.set_geom_size_env(gid, torch.from_numpy(z["gs"]).cuda().contiguous())
    st.set_geom_type_env(torch.from_numpy(z["gt"]).cuda().contiguous())
    if z["bm"].size: st.set_body_mass_env(bid, torch.from_numpy(z["bm"]).cuda().contiguous())
    ctrl = torch.from_numpy(z["ctrl"]).cuda().contiguous()
d_ = E.Derived(hm, n, ["qacc", "nefc", "solver_niter"])
E.forward(hm, st, ctrl, d_)
torch.cuda.synchronize()
A = {k: getattr(st, k).cpu().numpy() for k in ("qpos", "qvel", "act", "qacc_warmstart")}
gs, gt = st.geom_size_env.cpu().numpy(), st.geom_type_env.cpu().numpy()
bm = st.body_mass_env.cpu().numpy() if st.body_mass_env is not None else None
c = ctrl.cpu().numpy()
ga, gn, gi = d_["qacc"].cpu().numpy().astype(np.float64), d_["nefc"].cpu().numpy(), d_["solver_niter"].cpu().numpy()
stat = st.status.cpu().numpy()
bad = []
for e in range(n):
    d = oracle_forward(A["qpos"][e], A["qvel"][e], A["act"][e], c[e], A["qacc_warmstart"][e], gs[e].astype(np.float64), gt[e], None if bm is None else bm[e], gid, bid)
    rel = np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max())
    if rel > 1e-3 or mode == "replay":
        bad.append(e)
        i = int(np.abs(ga[e] - d.qacc).argmax())
        print(f"env {e}: rel {rel:.3e} nefc gpu {gn[e]} oracle {d.nefc} ncon {d.ncon} niter gpu {gi[e]} oracle {d.solver_niter} status {stat[e]} warn {d.warn} worst dof {i} gpu {ga[e][i]:.4f} oracle {d.qacc[i]:.4f} |qacc|max {np.abs(d.qacc).max():.2f} geomtype {gt[e]}")
        print("   oracle efc types", list(d.efc_type[:d.nefc]), "pairs", list(d.con_pair[:d.ncon]))
if mode == "scan":
    print("bad envs", bad)
    if bad:
        np.savez(fn, **{k: v[bad] for k, v in A.items()}, gs=gs[bad], gt=gt[bad], bm=(bm[bad] if bm is not None else np.zeros(0, np.float32)), ctrl=c[bad], gid=gid, bid=bid)

if mode == "replay":
    # stage by stage for the dumped states: efc_D / efc_aref rows, smooth force, unconstrained acceleration
    dump = E.debug_dump(hm, st, ctrl).cpu().numpy()
    for e in range(n):
        d = oracle_forward(A["qpos"][e], A["qvel"][e], A["act"][e], c[e], A["qacc_warmstart"][e], gs[e].astype(np.float64), gt[e], None if bm is None else bm[e], gid, bid)
        ne = d.nefc
        for nm, ref in (("efc_D", d.efc_D[:ne]), ("efc_aref", d.efc_aref[:ne]), ("smooth", d.qfrc_smooth), ("qaccsm", d.qacc_smooth), ("qacc", d.qacc), ("qfrccon", d.qfrc_constraint)):
            off = hm.layout(nm); got = dump[e, off:off + ref.size]
            print(f"  env {e} {nm}: max abs diff {np.abs(got - ref).max():.3e} (ref max {np.abs(ref).max():.3e})")
            if nm in ("efc_D", "efc_aref"):
                print("     gpu", np.array2string(got[5:], precision=4, max_line_width=250)); print("     ora", np.array2string(np.asarray(ref)[5:], precision=4, max_line_width=250))
        print("   con dist", d.con_dist[:d.ncon], "pos", d.con_pos[:d.ncon], "frame n", d.con_frame[:d.ncon, :3])
        print("   scal", dump[e, hm.layout("scal"):hm.layout("scal") + 12])
