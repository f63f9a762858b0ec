"""Reorient envs (myoHandReorient8/100-v0): reference-pinned env arithmetic and convex narrow phase (CPU), HIP-vs-oracle (GPU)."""
import os

import numpy as np
import pytest

from myosuite_amd.model import synth
from oracle import env_oracle as EO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RK = ("pos_align", "rot_align", "act_reg", "drop", "bonus", "sparse", "solved", "done", "dense")
WT = {"pos_align": 1.0, "rot_align": 1.0, "act_reg": 5.0, "drop": 5.0, "bonus": 10.0}



_ERR = {}


def _record_err(nerr):
    """worst teacher-forced observation error by index, per test (printed at exit with -s; evidence for the tolerances)"""
    import inspect
    name = inspect.stack()[1].function
    cur = _ERR.get(name)
    _ERR[name] = nerr.copy() if cur is None or cur.shape != nerr.shape else np.maximum(cur, nerr)
    import atexit
    if not getattr(_record_err, "_hooked", False):
        _record_err._hooked = True
        atexit.register(lambda: [print("TEACHER-FORCED-ERR", k, len(v), f"max {v.max():.3e} at {int(v.argmax())}", "top", np.sort(v)[-5:][::-1].round(7).tolist()) for k, v in _ERR.items()])

def test_reorient_oracle_arithmetic_matches_reference_vectors():
    g = np.load(os.path.join(G, "ref_reorient_env.npz"))
    n = g["qpos"].shape[0]
    pen, tar = float(g["pen_length"]), float(g["tar_length"])
    seen = {"drop": 0, "bonus": 0}
    for i in range(n):
        # the oracle takes (R_obj, axis_half): feed a rotation whose third column reproduces the golden top-bot vector
        d1 = g["top_minus_bot"][i]; ah = 0.5 * np.linalg.norm(d1)
        z = d1 / np.linalg.norm(d1)
        x = np.cross(z, [0.3, 0.5, 0.8]); x /= np.linalg.norm(x); y = np.cross(z, x)
        R = np.stack([x, y, z], 1)
        obs, rwd = EO.reorient_obs_reward(g["qpos"][i], g["qvel"][i], g["act"][i], g["obj_xpos"][i], R, g["eps_pos"][i], ah,
                                          g["ttop_minus_tbot"][i] / tar, g["actuator_length"][i], g["actuator_velocity"][i],
                                          g["actuator_force"][i], float(g["dt"]), pen, WT)
        assert obs.shape == (200,)
        np.testing.assert_allclose(obs, g["obs"][i], rtol=2e-6, atol=2e-6)
        for k in RK:
            np.testing.assert_allclose(float(rwd[k]), g[f"rwd_{k}"][i], rtol=1e-9, atol=1e-9, err_msg=k)
        seen["drop"] += int(rwd["done"]); seen["bonus"] += int(rwd["bonus"] > 0)
    assert 0 < seen["drop"] < n and seen["bonus"] > 0
    for e, q, m in zip(g["euler"], g["euler2quat"], g["quat2mat"]):
        np.testing.assert_allclose(EO.euler2quat(e), q, atol=1e-14)
        w, x, y, z = q
        np.testing.assert_allclose([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], m[:, 2], atol=1e-14)


def test_reorient_model_and_registry():
    from myosuite_amd.envs import registry
    cm = synth.get_model("hand_reorient")
    assert (cm.nq, cm.nv, cm.nu) == (29, 29, 39) and cm.njmax <= 64 and cm.npair == 20
    jn = list(cm.names["joint"].keys())
    assert jn[-6:] == ["OBJTx", "OBJTy", "OBJTz", "OBJRx", "OBJRy", "OBJRz"]                # myohand_sar.xml:27-32
    for vid in ("myoHandReorient8-v0", "myoHandReorient100-v0", "myoFatiHandReorient100-v0", "myoReafHandReorient8-v0"):
        s = registry.spec(vid)
        assert s["max_episode_steps"] == 50 and s["kwargs"]["frame_skip"] == 5
    assert len(synth.REORIENT_CAPS_100) == 25 and len(synth.REORIENT_CAPS_8) == 2
    for vid in ("myoHandReorientID-v0", "myoHandReorientOOD-v0"):
        assert registry.spec(vid)["max_episode_steps"] == 50
        assert synth.reorient_tables(registry.spec(vid)["kwargs"]["geometries"]).shape == (4, 250, 3)


def test_segment_vs_convex_signed_distance_against_sampling(oracle_lib):
    """oracle/mmo_collision.inc seg_shape (capsule axis vs box / cylinder / ellipsoid) against dense sampling of the
    closed-form signed distance along the segment."""
    from oracle import oracle as O
    rng = np.random.default_rng(0)
    th = np.linspace(0, np.pi, 240); ph = np.linspace(0, 2 * np.pi, 480)
    T, Ph = np.meshgrid(th, ph, indexing="ij")
    for gtype in (6, 5, 4):
        for trial in range(24):
            size = rng.uniform(0.01, 0.045, 3)
            a = rng.standard_normal(3); a *= rng.uniform(0.01, 0.09) / np.linalg.norm(a)
            u = rng.standard_normal(3)
            if trial % 4 == 0:
                u[2] = 0                                   # axis parallel to a face / to the cylinder caps
            u /= np.linalg.norm(u)
            h = rng.uniform(0.01, 0.03)
            sd, t, g = O.seg_shape(gtype, size, a, u, h)
            assert abs(np.linalg.norm(g) - 1) < 1e-9 and -h - 1e-12 <= t <= h + 1e-12
            ts = np.linspace(-h, h, 2001 if gtype != 4 else 121)
            P = a[None] + ts[:, None] * u[None]
            if gtype == 6:
                d = np.abs(P) - size
                ref = (np.linalg.norm(np.maximum(d, 0), axis=1) + np.minimum(d.max(axis=1), 0)).min()
            elif gtype == 5:
                dr = np.linalg.norm(P[:, :2], axis=1) - size[0]; dz = np.abs(P[:, 2]) - size[1]
                ref = np.where((dr > 0) & (dz > 0), np.hypot(np.maximum(dr, 0), np.maximum(dz, 0)), np.maximum(dr, dz)).min()
            else:
                S = np.stack([size[0] * np.sin(T) * np.cos(Ph), size[1] * np.sin(T) * np.sin(Ph), size[2] * np.cos(T)], -1).reshape(-1, 3)
                dist = np.array([np.linalg.norm(S - p, axis=1).min() for p in P])
                ref = np.where(((P / size) ** 2).sum(1) < 1, -dist, dist).min()
            assert abs(sd - ref) < (2e-5 if gtype != 4 else 2e-4), (gtype, trial, sd, ref)


@pytest.mark.gpu
def test_gpu_reorient_env_matches_oracle_env(oracle_lib):
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry
    cm = synth.get_model("hand_reorient")
    n, nsteps = 16, 8
    env = registry.make("myoHandReorient100-v0", num_envs=n, seed=9, autoreset=False)
    obs0, _ = env.reset(seed=9)
    assert obs0.shape == (n, 200)
    ep = env.episode.cpu().numpy()
    orc = []
    sizes = set(); types = set()
    for e in range(n):
        gt, size, ah, des = EO.reorient_reset_draws(env.size_tables_np, e, int(ep[e]) - 1, 9, env.tar_length)
        assert int(env.geom_type[e]) == gt
        types.add(gt)
        np.testing.assert_allclose(env.geom_size[e].cpu().numpy(), size, atol=1e-7)
        np.testing.assert_allclose(float(env.axis_half[e]), ah, rtol=1e-6)
        np.testing.assert_allclose(env.des_rot[e].cpu().numpy(), des, atol=2e-6)
        sizes.add(tuple(np.round(size, 4)))
        w = EO.ReorientEnvOracle(cm)
        o = w.reset(size, ah, env.des_rot[e].cpu().numpy().astype(np.float64), gt)
        np.testing.assert_allclose(obs0[e].cpu().numpy(), o, rtol=1e-4, atol=3e-5)
        orc.append(w)
    assert len(sizes) >= 3 and len(types) >= 3, types
    a = torch.empty(n, cm.nu, device="cuda")
    for s in range(nsteps):
        # teacher forcing (contact onsets are discontinuities of time-stepped dynamics)
        st = env.get_env_state()
        for e in range(n):
            d = orc[e].d
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                v = getattr(d, k).astype(np.float32); getattr(d, k)[:] = v
                st[k][e] = torch.from_numpy(v)
        env.set_env_state(st)
        E.uniform(a, 13, s)
        act = (0.3 + 0.5 * a).contiguous()
        obs, r, term, trunc, info = env.step(act)
        an = act.cpu().numpy()
        for e in range(n):
            o, dense, done, rd = orc[e].step(an[e].astype(np.float64))
            got = obs[e].cpu().numpy()
            tol = np.full(200, 5e-4)   # measured worst over the run 3.9e-5 (a muscle-force entry); round 2 allowed mforce 0.5, mvel / obj_vel 2e-2
            scale = np.maximum(1.0, np.abs(o))
            _record_err(np.abs(got - o) / scale)
            bad = np.abs(got - o) / scale > tol
            assert not bad.any(), (s, e, np.nonzero(bad)[0][:5], (np.abs(got - o) / scale)[bad][:5])
            for i, k in enumerate(E.RWD_KEYS_REORIENT):
                ref = float(rd[k])
                assert abs(float(env.rwd[e, i]) - ref) < 5e-4 * max(1.0, abs(ref)), (k, s, e)
            assert bool(term[e]) == done
    assert list(info["rwd_dict"].keys()) == E.RWD_KEYS_REORIENT


@pytest.mark.gpu
def test_gpu_reorient_drop_terminates_and_autoresets(oracle_lib):
    import torch
    from myosuite_amd.envs import registry
    env = registry.make("myoHandReorient8-v0", num_envs=64, seed=2)
    env.reset(seed=2)
    a = torch.zeros(64, env.cm.nu, device="cuda")        # relaxed hand: the 1.2 kg object slides off -> dropped
    dropped = torch.zeros(64, dtype=torch.bool, device="cuda")
    for _ in range(50):
        obs, r, term, trunc, info = env.step(a)
        dropped |= term
        assert torch.isfinite(obs).all() and torch.isfinite(r).all()
    assert float(dropped.float().mean()) > 0.5
    assert int((env.state.status & 0xA).max()) == 0       # no row overflow / no solver failure (bit0 = gimbal auto-reset is legal)


# ----------------------------------------------------------------------------------- PenTwirl (pen_v0.py)
def test_pen_oracle_arithmetic_matches_reference_vectors():
    g = np.load(os.path.join(G, "ref_pen_env.npz"))
    n = g["qpos"].shape[0]
    assert list(g["keys"]) == ["hand_jnt", "obj_pos", "obj_vel", "obj_rot", "obj_des_rot", "obj_err_pos", "obj_err_rot", "act"]
    z39 = np.zeros(39)
    for i in range(n):
        d1 = g["top_minus_bot"][i]
        z = d1 / np.linalg.norm(d1)
        x = np.cross(z, [0.3, 0.5, 0.8]); x /= np.linalg.norm(x); y = np.cross(z, x)
        obs, rwd = EO.reorient_obs_reward(g["qpos"][i], g["qvel"][i], g["act"][i], g["obj_xpos"][i], np.stack([x, y, z], 1),
                                          g["eps_pos"][i], 0.5 * np.linalg.norm(d1), g["ttop_minus_tbot"][i] / 0.13, z39, z39, z39,
                                          float(g["dt"]), 0.13, WT, obs_muscle=False)
        assert obs.shape == (83,)
        np.testing.assert_allclose(obs, g["obs"][i], rtol=2e-6, atol=2e-6)
        for k in RK:
            np.testing.assert_allclose(float(rwd[k]), g[f"rwd_{k}"][i], rtol=1e-9, atol=1e-9, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["myoHandPenTwirlRandom-v0", "myoHandPenTwirlFixed-v0"])
def test_gpu_pen_twirl_env_matches_oracle_env(oracle_lib, env_id):
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry
    cm = synth.get_model("hand_pen")
    n, nsteps = 8, 6
    env = registry.make(env_id, num_envs=n, seed=4, autoreset=False)
    obs0, _ = env.reset(seed=4)
    assert obs0.shape == (n, 83) and list(env.obs_dict.keys())[1:8] == ["hand_jnt", "obj_pos", "obj_vel", "obj_rot", "obj_des_rot",
                                                                       "obj_err_pos", "obj_err_rot"]
    ep = env.episode.cpu().numpy()
    orc = []
    for e in range(n):
        rng = (-1.0, 1.0, -1.0, 1.0) if "Random" in env_id else (0.0, 0.0, 0.0, 0.0)
        des = EO.pen_reset_draws(e, int(ep[e]) - 1, 4, 0.065, env.tar_length, rng)
        np.testing.assert_allclose(env.des_rot[e].cpu().numpy(), des, atol=2e-6)
        w = EO.PenTwirlEnvOracle(cm)
        o = w.reset(env.des_rot[e].cpu().numpy().astype(np.float64))
        np.testing.assert_allclose(obs0[e].cpu().numpy(), o, rtol=1e-4, atol=3e-5)
        orc.append(w)
    if "Fixed" in env_id:
        np.testing.assert_allclose(env.des_rot.cpu().numpy(), np.tile([0, 0, 1.0], (n, 1)), atol=1e-7)
    a = torch.empty(n, cm.nu, device="cuda")
    for s in range(nsteps):
        st = env.get_env_state()
        for e in range(n):
            d = orc[e].d
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                v = getattr(d, k).astype(np.float32); getattr(d, k)[:] = v
                st[k][e] = torch.from_numpy(v)
        env.set_env_state(st)
        E.uniform(a, 21, s)
        act = (0.3 + 0.5 * a).contiguous()
        obs, r, term, trunc, info = env.step(act)
        an = act.cpu().numpy()
        for e in range(n):
            o, dense, done, rd = orc[e].step(an[e].astype(np.float64))
            got = obs[e].cpu().numpy()
            tol = np.full(83, 5e-5)   # measured worst 1.6e-6
            _record_err(np.abs(got - o) / np.maximum(1.0, np.abs(o)))
            bad = np.abs(got - o) / np.maximum(1.0, np.abs(o)) > tol
            assert not bad.any(), (s, e, np.nonzero(bad)[0][:5])
            for i, k in enumerate(E.RWD_KEYS_REORIENT):
                ref = float(rd[k])
                assert abs(float(env.rwd[e, i]) - ref) < 5e-4 * max(1.0, abs(ref)), (k, s, e)
            assert bool(term[e]) == done
