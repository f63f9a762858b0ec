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
tc = torch.from_numpy(ctrl[7:8]).cuda()
hm32 = E.HipModel(cm, lanes_per_env=32); hm64 = E.HipModel(cm, lanes_per_env=64)
st = E.BatchState(hm32, 1)
st.qpos.copy_(torch.from_numpy(g["qpos"][s, 7:8].astype(np.float32))); st.qvel.copy_(torch.from_numpy(g["qvel"][s, 7:8].astype(np.float32))); st.act.copy_(torch.from_numpy(g["act"][s, 7:8].astype(np.float32)))
E.step(hm32, st, tc, 9)
pre = [x.clone() for x in (st.qpos, st.qvel, st.act, st.qacc_warmstart)]
def gpu_step(hm):
    s2 = E.BatchState(hm, 1)
    for dst, src in zip((s2.qpos, s2.qvel, s2.act, s2.qacc_warmstart), pre): dst.copy_(src)
    E.step(hm, s2, tc, 1)
    return s2.qvel.cpu().numpy()[0], s2.qacc_warmstart.cpu().numpy()[0]
v32, w32 = gpu_step(hm32); v64, w64 = gpu_step(hm64)
d = O.OracleData(om)
d.qpos[:] = pre[0][0].cpu().numpy(); d.qvel[:] = pre[1][0].cpu().numpy(); d.act[:] = pre[2][0].cpu().numpy(); d.qacc_warmstart[:] = pre[3][0].cpu().numpy(); d.ctrl[:] = ctrl[7]
d.step(1)
print("oracle nefc", d.nefc, "niter", d.solver_niter)
print("dv32", np.abs(v32 - d.qvel).max(), "dv64", np.abs(v64 - d.qvel).max())
print("ora qacc", np.array(d.qacc_warmstart)); print("g32 qacc", w32); print("g64 qacc", w64)
# single-forward dumps from the same pre-state
for hm, nm in ((hm32, "g32"), (hm64, "g64")):
    s2 = E.BatchState(hm, 1)
    for dst, src in zip((s2.qpos, s2.qvel, s2.act, s2.qacc_warmstart), pre): dst.copy_(src)
    dump = E.debug_dump(hm, s2, tc).cpu().numpy()[0]
    o = hm.layout("qacc"); print(nm, "fwd qacc", dump[o:o+cm.nv]); o = hm.layout("efc_active"); print("   active", np.nonzero(dump[o:o+64])[0], "niter", dump[hm.layout("scal")])
    o = hm.layout("qfrccon"); print("   qfrccon", dump[o:o+cm.nv])
print("ora qfrccon", np.array(d.qfrc_constraint)); print("ora rows pos", d.efc_pos[:d.nefc])
print("---- variants")
for hm, nm in ((hm32, "g32"), (hm64, "g64")):
    for zero_warm in (0, 1):
        s2 = E.BatchState(hm, 1)
        for dst, src in zip((s2.qpos, s2.qvel, s2.act, s2.qacc_warmstart), pre): dst.copy_(src)
        if zero_warm: s2.qacc_warmstart.zero_()
        dump = E.debug_dump(hm, s2, tc).cpu().numpy()[0]
        o = hm.layout("qacc"); qa = dump[o:o+cm.nv]; o = hm.layout("qaccsm"); qs = dump[o:o+cm.nv]
        print(nm, "zero_warm", zero_warm, "qacc[16:]", qa[16:], "scal [niter chose_ws cost_ws cost_sm alpha0..3]", dump[hm.layout("scal"):hm.layout("scal")+8])
print("warm[16:]", pre[3][0].cpu().numpy()[16:])
