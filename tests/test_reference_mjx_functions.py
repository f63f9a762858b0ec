"""The reference's engine-level test file, restated against this engine: myosuite/tests/test_mjx.py (TestMjxFunctions) loads the
myoFinger model, puts it on the device, makes data, takes one step, and compares one forward pass with libmujoco's `mj_forward`
(xpos / xquat to 1e-5, qpos / qvel untouched).  Same four tests, same model family (the synthetic `finger`), same
tolerances; `mjx.put_model / make_data / step / forward` are `HipModel / BatchState / mm_step / mm_forward` behind the C ABI and the
role libmujoco plays there is played by the fp64 oracle (parity unpinned).  A batch of 64 states instead of one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 64


@pytest.fixture(scope="module")
def finger(oracle_lib):
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.model import synth
    cm = synth.get_model("finger")
    return dict(E=E, O=oracle_lib, torch=torch, cm=cm, hm=E.HipModel(cm), om=oracle_lib.OracleModel(cm))


def test_model_loading(finger):                        # test_mjx.py:36-42
    cm, hm = finger["cm"], finger["hm"]
    assert hm is not None
    assert (hm.cm.nq, hm.cm.nv, hm.cm.nu) == (cm.nq, cm.nv, cm.nu) and cm.nq == 4 and cm.nu == 5     # myofinger_v0: 4 joints, 5 muscles
    assert hm.info(finger["E"].INFO_KERNEL_FAMILY) >= 0 and hm.launch_lanes(N) in (4, 8, 16, 32, 64)


def test_data_creation(finger):                        # test_mjx.py:44-51
    E, cm = finger["E"], finger["cm"]
    st = E.BatchState(finger["hm"], N)
    assert st.qpos.shape == (N, cm.nq) and st.qvel.shape == (N, cm.nv) and st.act.shape == (N, cm.nu)
    assert np.array_equal(st.qpos[0].cpu().numpy(), cm.qpos0.astype(np.float32)) and float(st.qvel.abs().max()) == 0.0


def test_step_simulation(finger):                      # test_mjx.py:53-80
    E, torch, cm = finger["E"], finger["torch"], finger["cm"]
    st = E.BatchState(finger["hm"], N)
    q0, v0 = st.qpos.clone(), st.qvel.clone()
    E.step(finger["hm"], st, torch.zeros(N, cm.nu, device="cuda"), 1)          # one mj_step with zero control input
    torch.cuda.synchronize()
    qpos_changed = not torch.allclose(q0, st.qpos, atol=1e-6)
    qvel_changed = not torch.allclose(v0, st.qvel, atol=1e-6)
    assert qpos_changed or qvel_changed, "qpos or qvel should change after a step with gravity"
    assert int(st.status.max()) == 0 and float(st.time[0]) == pytest.approx(cm.timestep)
    d = finger["O"].OracleData(finger["om"]); d.step(1)                        # and it is the oracle's step
    np.testing.assert_allclose(st.qpos[0].cpu().numpy(), d.qpos, atol=1e-6)
    np.testing.assert_allclose(st.qvel[0].cpu().numpy(), d.qvel, atol=1e-4)


def test_forward_kinematics(finger):                   # test_mjx.py:82-140
    E, O, torch, cm = finger["E"], finger["O"], finger["torch"], finger["cm"]
    hm = finger["hm"]
    st = E.BatchState(hm, N)
    rng = np.random.default_rng(0)
    lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
    q = (lo + (hi - lo) * rng.random((N, cm.nq))).astype(np.float32); q[0] = cm.qpos0                # env 0 = the reference's case (fresh data)
    v = (0.5 * rng.standard_normal((N, cm.nv))).astype(np.float32); v[0] = 0
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
    dv = E.Derived(hm, N, ["xpos", "xquat", "subtree_com", "site_xpos"])
    E.forward(hm, st, None, dv)
    torch.cuda.synchronize()
    xpos, xquat, com = dv["xpos"].cpu().numpy(), dv["xquat"].cpu().numpy(), dv["subtree_com"].cpu().numpy()
    assert np.any(xpos != 0.0), "xpos should not be all zeros after forward kinematics"
    assert np.all(np.isfinite(xquat)) and np.all(np.isfinite(com))
    for e in range(N):
        d = O.OracleData(finger["om"]); d.qpos[:] = q[e]; d.qvel[:] = v[e]
        d.forward()
        np.testing.assert_allclose(xpos[e], d.xpos, atol=1e-5, err_msg="mm_forward xpos does not match the oracle's mj_forward xpos")
        sgn = np.sign(np.sum(xquat[e] * d.xquat, axis=1, keepdims=True)); sgn[sgn == 0] = 1       # q and -q are one orientation
        np.testing.assert_allclose(xquat[e] * sgn, d.xquat, atol=1e-5, err_msg="mm_forward xquat does not match the oracle's")
        # mm_derived.subtree_com holds the COM of the kinematic TREE a body belongs to (the point MuJoCo's cdof / cinert refer to,
        # include/myosim.h): equal to mjData.subtree_com at the tree roots; the reference only asks for it to exist
        root = np.asarray(cm.arrays["BODY_ROOTID"])
        np.testing.assert_allclose(com[e][1:], d.subtree_com[root[1:]], atol=1e-5)
        np.testing.assert_allclose(dv["site_xpos"][e].cpu().numpy(), d.site_xpos, atol=1e-5)
    # forward must not change the state
    np.testing.assert_array_equal(st.qpos.cpu().numpy(), q, err_msg="mm_forward should not change qpos")
    np.testing.assert_array_equal(st.qvel.cpu().numpy(), v, err_msg="mm_forward should not change qvel")
    assert float(st.time.abs().max()) == 0.0
