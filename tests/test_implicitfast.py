"""mjINT_IMPLICITFAST (the reference's MJX base env names it: envs/myo/mjx/mjx_base_env.py:54-55) in the
oracle and in the fused kernel: implicit-in-velocity Euler with the velocity derivative of the passive and actuator forces
(joint / tendon damping, muscle force-velocity slope, affine velocity servos), Coriolis terms dropped, matrix symmetric."""
import numpy as np
import pytest

from myosuite_amd.model import mjcf, synth
from oracle import oracle as O


def _elbow(integrator, vmax=1.5, dt=0.002):
    s = synth.make_elbow(); s.integrator = integrator; s.timestep = dt
    for a in s.actuators:
        g = list(a.gainprm); g[6] = vmax; a.gainprm = tuple(g)
    return s.compile()


def _run(cm, T, act=0.8, q0=1.0):
    d = O.OracleData(O.OracleModel(cm)); d.qpos[0] = q0; d.act[:] = act
    for _ in range(int(round(T / cm.timestep))):
        d.ctrl[:] = act; d.step()
    return float(d.qpos[0])


def test_without_velocity_dependent_actuation_implicitfast_is_euler_with_implicit_damping(oracle_lib):
    """act = 0: the only velocity-dependent smooth force is joint damping, and M + h*diag(damping) is what eulerdamp solves."""
    d0 = O.OracleData(O.OracleModel(_elbow(0))); d1 = O.OracleData(O.OracleModel(_elbow(3)))
    for d in (d0, d1):
        d.qpos[0] = 1.0; d.qvel[0] = 2.0
    for _ in range(300):
        d0.step(); d1.step()
    assert abs(d0.qpos[0] - d1.qpos[0]) < 1e-12 and abs(d0.qvel[0] - d1.qvel[0]) < 1e-12


def test_stiff_force_velocity_curve_needs_the_implicit_integrator(oracle_lib):
    """vmax = 0.1 L0/s makes the muscles' force-velocity slope stiff (dt * |M^-1 df/dv| >> 2): against a small-step RK4 reference
    explicit Euler at the task timestep is off by half a radian after one second, implicitfast by microradians, and its error
    halves with the timestep (first order)."""
    ref = _run(_elbow(1, 0.1, 2e-5), 1.0)
    e_euler = abs(_run(_elbow(0, 0.1, 0.002), 1.0) - ref)
    e_impl = [abs(_run(_elbow(3, 0.1, dt), 1.0) - ref) for dt in (0.002, 0.001, 0.0005)]
    assert e_euler > 0.1 and e_impl[0] < 2e-5
    assert 1.6 < e_impl[0] / e_impl[1] < 2.6 and 1.6 < e_impl[1] / e_impl[2] < 2.6


def test_velocity_derivative_matches_finite_differences_of_the_smooth_force(oracle_lib):
    """(M - h D) with D = d qfrc_smooth / d qvel minus its Coriolis part: on a model at rest velocity (v = 0) the Coriolis
    derivative vanishes, so one implicitfast step must equal the step built from a finite-difference D of the oracle's own
    qfrc_smooth -- the hand: 39 muscles with tendon transmission, joint damping."""
    s = synth.make_hand(); s.integrator = 3
    cm = s.compile(); om = O.OracleModel(cm)
    rng = np.random.default_rng(1)
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    q = lo + 0.2 * (hi - lo) + 0.6 * (hi - lo) * rng.random(cm.nq)
    act = 0.2 + 0.6 * rng.random(cm.na)

    def smooth(v):
        d = O.OracleData(om); d.qpos[:] = q; d.qvel[:] = v; d.act[:] = act; d.ctrl[:] = act
        d.forward()
        return d.qfrc_smooth.copy(), d
    f0, d0 = smooth(np.zeros(cm.nv))
    assert d0.nefc == 0
    eps = 1e-6
    D = np.zeros((cm.nv, cm.nv))
    for k in range(cm.nv):
        e = np.zeros(cm.nv); e[k] = eps
        D[:, k] = (smooth(e)[0] - smooth(-e)[0]) / (2 * eps)
    D = 0.5 * (D + D.T)
    # chain mask (the pattern of M): the hand's tendons stay on one chain, the mask only removes rounding-level entries
    M = d0.full_M(); h = cm.timestep
    qacc = np.linalg.solve(M - h * np.where(M != 0, D, 0.0), f0)
    d1 = O.OracleData(om); d1.qpos[:] = q; d1.act[:] = act; d1.ctrl[:] = act
    d1.step()
    np.testing.assert_allclose(d1.qvel, h * qacc, rtol=2e-5, atol=1e-8)


def test_leg_implicit_model_and_mjcf_round_trip(oracle_lib):
    cm = synth.get_model("leg_implicit"); base = synth.get_model("leg")
    assert int(cm.arrays["OPT_I"][16]) == 3 and (cm.nq, cm.nv, cm.nu) == (base.nq, base.nv, base.nu)
    gp = cm.arrays["ACT_GAINPRM"].reshape(-1, 9)
    assert np.all(gp[:, 6] == 1.5) and np.all(base.arrays["ACT_GAINPRM"].reshape(-1, 9)[:, 6] == 10.0)      # MuJoCo's default vmax
    cm2 = mjcf.load(mjcf.dump(synth.make_leg(implicit=True))).compile()
    assert int(cm2.arrays["OPT_I"][16]) == 3
    da, db = O.OracleData(O.OracleModel(cm)), O.OracleData(O.OracleModel(cm2))
    for x in (da, db):
        x.qpos[:] = cm.key_qpos[2]; x.act[:] = 0.5; x.ctrl[:] = 0.5
        x.step(20)
    assert np.abs(da.qpos - db.qpos).max() == 0.0                      # the MJCF round trip carries the integrator
    d = O.OracleData(O.OracleModel(cm)); d.qpos[:] = cm.key_qpos[2]; d.qvel[:] = cm.key_qvel[2]; d.act[:] = 1.0
    for _ in range(50):
        d.ctrl[:] = 1.0; d.step(10)                      # full co-activation of all 80 muscles
    assert d.warn == 0 and np.all(np.isfinite(d.qpos)) and np.abs(d.qvel).max() < 50


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["elbow", "hand", "leg_implicit"])
def test_hip_implicitfast_matches_oracle(oracle_lib, name):
    import torch
    from myosuite_amd import engine as E
    if name == "leg_implicit":
        cm = synth.get_model(name)
    else:
        s = synth.builders()[name](); s.integrator = 3
        cm = s.compile()
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    n = 12
    rng = np.random.default_rng(3)
    if name == "leg_implicit":
        q = np.tile(cm.key_qpos[2].astype(np.float64), (n, 1)); q[:, 7:] += rng.uniform(-0.05, 0.05, (n, cm.nq - 7))
        v = rng.standard_normal((n, cm.nv)) * 0.2
    else:
        lo, hi = cm.jnt_range[:, 0].astype(np.float64), cm.jnt_range[:, 1].astype(np.float64)
        q = lo + (hi - lo) * rng.random((n, cm.nq)); v = rng.standard_normal((n, cm.nv))
    act = rng.random((n, cm.na)); ctrl = rng.random((n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q.astype(np.float32))); st.qvel.copy_(torch.from_numpy(v.astype(np.float32)))
    st.act.copy_(torch.from_numpy(act.astype(np.float32)))
    ds = []
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e].astype(np.float32); d.qvel[:] = v[e].astype(np.float32); d.act[:] = act[e].astype(np.float32)
        d.ctrl[:] = ctrl[e]; ds.append(d)
    c = torch.from_numpy(ctrl).cuda()
    # one substep: teacher-forced
    E.step(hm, st, c, 1)
    for d in ds:
        d.step(1)
    qo = np.array([d.qpos for d in ds]); vo = np.array([d.qvel for d in ds])
    assert np.abs(st.qpos.cpu().numpy() - qo).max() < 2e-6 and np.abs(st.qvel.cpu().numpy() - vo).max() < 5e-4 * max(1.0, np.abs(vo).max())
    # 40 substeps free running
    E.step(hm, st, c, 40)
    for d in ds:
        d.step(40)
    qo = np.array([d.qpos for d in ds])
    err = np.abs(st.qpos.cpu().numpy() - qo).max(axis=1)
    assert np.median(err) < 5e-5 and err.max() < (5e-3 if name == "leg_implicit" else 5e-4), (np.median(err), err.max())
    assert int(st.status.max()) == 0 and max(d.warn for d in ds) == 0
