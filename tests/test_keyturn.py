"""KeyTurn envs (myoHandKeyTurn{Fixed,Random}-v0): reference-pinned env arithmetic (CPU), HIP-vs-oracle (GPU)."""
import os

import numpy as np
import pytest

from myosuite_amd.model import synth
from oracle import env_oracle as EO

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RK = ("key_turn", "IFtip_approach", "THtip_approach", "act_reg", "bonus", "penalty", "sparse", "solved", "done", "dense")


def test_keyturn_oracle_arithmetic_matches_reference_vectors():
    g = np.load(os.path.join(G, "ref_keyturn_env.npz"))
    for tag in ("a", "b"):
        seen = {"done": 0, "solved": 0}
        for i in range(g["qpos"].shape[0]):
            obs, rwd = EO.keyturn_obs_reward(g["qpos"][i], g["qvel"][i], g["act"][i], g["keyhead"][i], g["iftip"][i], g["thtip"][i],
                                             float(g["dt"]), float(g[f"{tag}_goal_th"]), EO.KeyTurnEnvOracle.RWD_KEYS_WT)
            assert obs.shape == (93,)
            np.testing.assert_allclose(obs, g[f"{tag}_obs"][i], rtol=2e-6, atol=2e-6)
            for k in RK:
                np.testing.assert_allclose(float(rwd[k]), g[f"{tag}_rwd_{k}"][i], rtol=1e-9, atol=1e-9, err_msg=k)
            seen["done"] += int(rwd["done"]); seen["solved"] += int(rwd["solved"])
        assert seen["done"] > 0 and seen["solved"] > 0


def test_keyturn_model_and_registry():
    from myosuite_amd.envs import registry
    cm = synth.get_model("hand_keyturn")
    assert (cm.nq, cm.nv, cm.nu) == (24, 24, 39)
    assert cm.names["body"]["key"] == cm.nbody - 1                      # key_turn_v0.py:164 addresses body_pos[-1]
    fl = cm.arrays["DOF_FRICTIONLOSS"]
    assert fl[-1] == np.float32(0.02) and not fl[:-1].any()             # myohand_keyturn.xml:29
    assert registry.spec("myoHandKeyTurnRandom-v0")["kwargs"]["goal_th"] == 2 * np.pi
    for vid in ("myoHandKeyTurnFixed-v0", "myoSarcHandKeyTurnRandom-v0", "myoReafHandKeyTurnFixed-v0"):
        assert registry.spec(vid)["max_episode_steps"] == 200


def test_keyturn_oracle_env_turns_the_key(oracle_lib):
    """Physics sanity of the friction-loss key: at rest it stays put (stick), a torque pulse through contact-free state
    integration decays (dry friction + damping dissipate)."""
    cm = synth.get_model("hand_keyturn")
    w = EO.KeyTurnEnvOracle(cm)
    w.reset(0.3)
    for _ in range(3):                       # the passive hand reaches the key a few env-steps later
        w.step(np.zeros(cm.nu))
        assert w.d.ncon == 0
    assert abs(w.d.qpos[-1] - 0.3) < 1e-3 and abs(w.d.qvel[-1]) < 5e-3 and w.d.nefc >= 1
    w.reset(0.3)
    w.d.qvel[-1] = 5.0
    w.d.step(1)
    v1 = abs(w.d.qvel[-1])
    w.d.step(1)
    assert v1 < 5.0 and abs(w.d.qvel[-1]) < v1


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", ["myoHandKeyTurnRandom-v0", "myoHandKeyTurnFixed-v0"])
def test_gpu_keyturn_env_matches_oracle_env(oracle_lib, env_id):
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry
    cm = synth.get_model("hand_keyturn")
    n, nsteps = 8, 6
    rnd = "Random" in env_id
    env = registry.make(env_id, num_envs=n, seed=6, autoreset=False)
    obs0, _ = env.reset(seed=6)
    assert obs0.shape == (n, 93)
    ep = env.episode.cpu().numpy()
    orc = []
    for e in range(n):
        lo, hi = env.key_init_range
        kq = np.float32(lo + (hi - lo) * EO.env_draw(1, e, int(ep[e]) - 1, 6, 17)[0])
        assert abs(float(env.key_q0[e, 0]) - float(kq)) < 1e-6
        kp = None
        if rnd:
            kp = (env.key_init_pos + (np.float32(-0.01) + np.float32(0.02) * EO.env_draw(3, e, int(ep[e]) - 1, 6, 18))).astype(np.float32)
            np.testing.assert_allclose(env.body_pos[e].cpu().numpy(), kp, atol=1e-7)
        w = EO.KeyTurnEnvOracle(cm, goal_th=env.goal_th)
        o = w.reset(float(kq), None if kp is None else kp.astype(np.float64))
        np.testing.assert_allclose(obs0[e].cpu().numpy(), o, rtol=1e-4, atol=3e-5)
        orc.append(w)
    if rnd:
        assert float((env.body_pos.max(0).values - env.body_pos.min(0).values).min()) > 1e-3
    a = torch.empty(n, cm.nu, device="cuda")
    for s in range(nsteps):
        st = env.get_env_state()
        for e in range(n):                       # teacher-forced per env-step (contact onsets are discontinuous)
            d = orc[e].d
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                v = getattr(d, k).astype(np.float32); getattr(d, k)[:] = v
                st[k][e] = torch.from_numpy(v)
        env.set_env_state(st)
        E.uniform(a, 29, s)
        act = (0.2 + 0.7 * a).contiguous()
        obs, r, term, trunc, info = env.step(act)
        an = act.cpu().numpy()
        for e in range(n):
            o, dense, done, rd = orc[e].step(an[e].astype(np.float64))
            got = obs[e].cpu().numpy()
            tol = np.full(93, 2e-3); tol[23:46] = 1e-2; tol[47] = 1e-2
            bad = np.abs(got - o) / np.maximum(1.0, np.abs(o)) > tol
            assert not bad.any(), (s, e, np.nonzero(bad)[0][:5], (np.abs(got - o))[bad][:5])
            for i, k in enumerate(E.RWD_KEYS_KEYTURN):
                ref = float(rd[k])
                assert abs(float(env.rwd[e, i]) - ref) < 5e-3 * max(1.0, abs(ref)), (k, s, e)
            assert bool(term[e]) == done
    assert list(info["rwd_dict"].keys()) == E.RWD_KEYS_KEYTURN
    assert int(env.state.status.max()) == 0
