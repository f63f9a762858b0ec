import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myosuite_amd.model import synth
from myosuite_amd import engine as E
from oracle import oracle as O
np.set_printoptions(precision=5, suppress=True, linewidth=220)
cm = synth.get_model("hand"); om = O.OracleModel(cm); hm = E.HipModel(cm)
rng = np.random.default_rng(0)
lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
nenv = 37
q0 = (lo + (hi - lo) * rng.random((nenv, cm.nq))).astype(np.float32)
ctrl = rng.random((nenv, cm.nu)).astype(np.float32)
st = E.BatchState(hm, nenv); st.qpos.copy_(torch.from_numpy(q0))
tc = torch.from_numpy(ctrl).cuda()
ds = [O.OracleData(om) for _ in range(nenv)]
for s in range(30):
    # teacher-force the ORACLE with the GPU state, then step both once
    gq = st.qpos.cpu().numpy(); gv = st.qvel.cpu().numpy(); ga = st.act.cpu().numpy(); gw = st.qacc_warmstart.cpu().numpy()
    for e, d in enumerate(ds):
        d.qpos[:] = gq[e]; d.qvel[:] = gv[e]; d.act[:] = ga[e]; d.qacc_warmstart[:] = gw[e]; d.ctrl[:] = ctrl[e]
        d.step(1)
    pre = [x.clone() for x in (st.qpos, st.qvel, st.act, st.qacc_warmstart)]
    E.step(hm, st, tc, 1)
    nv_ = st.qvel.cpu().numpy(); ov = np.stack([d.qvel for d in ds]); nw = st.qacc_warmstart.cpu().numpy(); ow = np.stack([d.qacc_warmstart for d in ds])
    err = np.abs(nv_ - ov).max(axis=1); e = int(np.argmax(err))
    print(s, "one-step dv max", err.max(), "env", e, "nefc", ds[e].nefc, "niter", ds[e].solver_niter, "dwarm", np.abs(nw[e]-ow[e]).max(), "max|qacc|", np.abs(ow[e]).max())
    if err.max() > 5e-3:
        d = ds[e]
        print(" gpu warm(qacc)", nw[e]); print(" ora qacc      ", ow[e]); print(" ora qaccsm", np.array(d.qacc_smooth))
        print(" ora efc: pos", d.efc_pos[:d.nefc], "D", d.efc_D[:d.nefc], "aref", d.efc_aref[:d.nefc], "frc", d.efc_force[:d.nefc])
        print(" ora qfrc_con", np.array(d.qfrc_constraint))
        st2 = E.BatchState(hm, nenv)
        for dst, src in zip((st2.qpos, st2.qvel, st2.act, st2.qacc_warmstart), pre): dst.copy_(src)
        dump = E.debug_dump(hm, st2, tc).cpu().numpy()
        o = hm.layout("scal"); print(" gpu scal [cost_ws cost_sm cost niter | alpha gn sn best_cost]*", dump[e, o:o+28])
        o = hm.layout("qacc"); print(" gpu fwd qacc", dump[e, o:o+cm.nv])
        o = hm.layout("qaccsm"); print(" gpu fwd qaccsm", dump[e, o:o+cm.nv])
        o = hm.layout("efc_active"); act = dump[e, o:o+64]; print(" gpu active lanes", np.nonzero(act)[0], "D", dump[e, hm.layout("efc_D"):hm.layout("efc_D")+64][act>0], "aref", dump[e, hm.layout("efc_aref"):hm.layout("efc_aref")+64][act>0])
        d2 = O.OracleData(om)
        d2.qpos[:] = pre[0][e].cpu().numpy(); d2.qvel[:] = pre[1][e].cpu().numpy(); d2.act[:] = pre[2][e].cpu().numpy(); d2.ctrl[:] = ctrl[e]
        d2.forward()
        for nm, on in (("tenlen", "ten_length"), ("tenvel", "ten_velocity"), ("actfrc", "actuator_force"), ("bias", "qfrc_bias"), ("smooth", "qfrc_smooth")):
            ref = getattr(d2, on); o = hm.layout(nm); got = dump[e, o:o+ref.size]
            k = int(np.argmax(np.abs(got-ref)))
            print(f" {nm}: max abs err {np.abs(got-ref).max():.3e} at index {k}: gpu {got[k]:.6f} ora {ref[k]:.6f}")
        tj = dump[e, hm.layout("tenj"):hm.layout("tenj") + cm.ntenJ]
        adr = cm.arrays["TENJ_ADR"]; dofs = cm.arrays["TENJ_DOF"]
        tv = dump[e, hm.layout("tenvel"):hm.layout("tenvel")+cm.ntendon]; t = int(np.argmax(np.abs(tv - d2.ten_velocity)))
        print(" worst tendon", t, list(cm.names["tendon"])[t], "dofs", dofs[adr[t]:adr[t+1]], "gpu J", tj[adr[t]:adr[t+1]], "ora J", d2.ten_J[t, dofs[adr[t]:adr[t+1]]])
        print(" gpu tenlen", dump[e, hm.layout("tenlen")+t], "ora", d2.ten_length[t], " qpos", pre[0][e].cpu().numpy())
        M = dump[e, hm.layout("M"):hm.layout("M") + cm.nv * cm.nv].reshape(cm.nv, cm.nv)
        print(" M err", np.abs(M - d2.full_M()).max(), "cond(M)", np.linalg.cond(d2.full_M()))
        print(" ora M^-1 smooth (np.solve with GPU M, GPU smooth)", np.linalg.solve(M.astype(np.float64), dump[e, hm.layout("smooth"):hm.layout("smooth")+cm.nv].astype(np.float64))[:8])
        break
