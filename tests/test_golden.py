"""Golden pins: (a) vectors produced by executing the reference's own importable Python
(tests/golden/make_golden_reference.py), (b) regression trajectories of our oracle."""
import os
import types

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_fatigue_oracle_matches_reference_code():
    from oracle.env_oracle import FatigueOracle
    g = np.load(os.path.join(G, "ref_fatigue.npz"))
    for tag in ("seq5", "rand39"):
        acts = g[f"{tag}_acts"]
        na = acts.shape[1]
        tau = g[f"{tag}_tau"]
        f = FatigueOracle(np.full(na, tau[0]), np.full(na, tau[1]), float(g[f"{tag}_dt"]), na)
        for i, a in enumerate(acts):
            MA, MR, MF = f.compute_act(a)
            np.testing.assert_allclose(MA, g[f"{tag}_MA"][i], rtol=1e-12, atol=1e-15)   # reference test uses rtol 1e-5
            np.testing.assert_allclose(MR, g[f"{tag}_MR"][i], rtol=1e-12, atol=1e-15)
            np.testing.assert_allclose(MF, g[f"{tag}_MF"][i], rtol=1e-12, atol=1e-15)
            assert abs((MA + MR + MF).max() - 1) < 1e-12                               # tests/mjx/test_fatigue.py:60-78


@pytest.mark.parametrize("tag,model", [("elbow", "elbow"), ("hand", "hand")])
def test_env_oracle_obs_reward_match_reference_code(oracle_lib, models, tag, model):
    from oracle.env_oracle import PoseEnvOracle
    g = np.load(os.path.join(G, "ref_pose_env.npz"))
    env = PoseEnvOracle(models[model], pose_thd=float(g[f"{tag}_pose_thd"]))
    assert abs(env.dt - float(g[f"{tag}_dt"])) < 1e-8     # model stores timestep as float32
    env.dt = float(g[f"{tag}_dt"])
    for i in range(g[f"{tag}_qpos"].shape[0]):
        env.d.qpos[:] = g[f"{tag}_qpos"][i]; env.d.qvel[:] = g[f"{tag}_qvel"][i]; env.d.act[:] = g[f"{tag}_act"][i]
        env.target_jnt_value = g[f"{tag}_target"][i]
        obs = env.get_obs()
        np.testing.assert_array_equal(obs, g[f"{tag}_obs"][i])              # float32 vectors, bit exact
        rd = env.get_reward_dict(env.obs_dict)
        for k in ("pose", "bonus", "penalty", "act_reg", "sparse", "solved", "done", "dense"):
            np.testing.assert_allclose(float(rd[k]), g[f"{tag}_rwd_{k}"][i], rtol=1e-12, atol=1e-12)


def test_torch_reward_mirror_matches_reference_code():
    """PoseEnvV0.get_reward_dict (torch, product host logic) against the reference's numpy output."""
    import torch
    from myosuite_amd.envs.pose_v0 import PoseEnvV0
    g = np.load(os.path.join(G, "ref_pose_env.npz"))
    for tag, na in (("elbow", 6), ("hand", 39)):
        fake = types.SimpleNamespace(pose_thd=float(g[f"{tag}_pose_thd"]), cm=types.SimpleNamespace(na=na),
                                     rwd_keys_wt=PoseEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS)
        od = {"pose_err": torch.from_numpy(g[f"{tag}_target"] - g[f"{tag}_qpos"]), "act": torch.from_numpy(g[f"{tag}_act"])}
        rd = PoseEnvV0.get_reward_dict(fake, od)
        for k in ("pose", "bonus", "penalty", "act_reg", "sparse", "dense"):
            np.testing.assert_allclose(rd[k].double().numpy(), g[f"{tag}_rwd_{k}"], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(rd["solved"].numpy(), g[f"{tag}_rwd_solved"] > 0.5)
        np.testing.assert_array_equal(rd["done"].numpy(), g[f"{tag}_rwd_done"] > 0.5)


@pytest.mark.parametrize("name", ["elbow", "hand"])
def test_oracle_trajectory_regression(oracle_lib, models, name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgo", os.path.join(G, "make_golden_oracle.py"))
    mgo = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgo)
    g = np.load(os.path.join(G, f"oracle_traj_{name}.npz"))
    assert str(g["model_hash"]) == models[name].hash(), "synthetic model changed: regenerate tests/golden"
    r = mgo.rollout(name)
    for k in ("qpos", "qvel", "act"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-9, atol=1e-11)


def test_oracle_leg_walk_trajectory_regression(oracle_lib):
    """Pins the oracle's contact / equality / free-joint path and WalkEnvOracle against the committed trajectory."""
    import importlib.util
    from myosuite_amd.model import synth
    spec = importlib.util.spec_from_file_location("mgo", os.path.join(G, "make_golden_oracle.py"))
    mgo = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgo)
    g = np.load(os.path.join(G, "oracle_traj_leg.npz"))
    assert str(g["model_hash"]) == synth.get_model("leg").hash(), "synthetic leg changed: regenerate tests/golden"
    r = mgo.rollout_leg()
    for k in ("qpos", "qvel", "act", "obs", "dense"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-8, atol=1e-10)


def test_oracle_collider_trajectory_regression(oracle_lib):
    """Pins the colliders (multiplicities incl. mjc_CapsuleBox's one-or-two contacts, plane-box / plane-cylinder up to four, parallel
    capsules) and the contact solve of the checker against the committed `plane_toy` trajectory: contact COUNT per substep exactly,
    positions to 1e-8 over 160 substeps of falling, landing and settling."""
    import importlib.util
    from myosuite_amd.model import synth
    spec = importlib.util.spec_from_file_location("mgo", os.path.join(G, "make_golden_oracle.py"))
    mgo = importlib.util.module_from_spec(spec); spec.loader.exec_module(mgo)
    g = np.load(os.path.join(G, "oracle_traj_plane_toy.npz"))
    assert str(g["model_hash"]) == synth.get_model("plane_toy").hash(), "plane_toy changed: regenerate tests/golden"
    r = mgo.rollout_plane_toy()
    np.testing.assert_array_equal(r["ncon"], g["ncon"])
    assert int(g["ncon"].max()) >= 8 and len(np.unique(g["ncon"])) >= 5
    for k in ("qpos", "qvel"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-8, atol=1e-9)


def test_registry_mirrors_reference_ids():
    from myosuite_amd.envs import registry
    ids = registry.registry_specs()
    for base in ("myoElbowPose1D6MRandom-v0", "myoElbowPose1D6MFixed-v0", "myoHandPoseRandom-v0", "myoHandPose0Fixed-v0"):
        assert base in ids
        assert base[:3] + "Sarc" + base[3:] in ids and base[:3] + "Fati" + base[3:] in ids
    assert "myoReafHandPoseRandom-v0" in ids and "myoReafElbowPose1D6MRandom-v0" not in ids
    s = ids["myoHandPoseRandom-v0"]
    assert s["max_episode_steps"] == 100 and s["kwargs"]["pose_thd"] == 0.7 and s["kwargs"]["reset_type"] == "random"
    assert ids["myoElbowPose1D6MRandom-v0"]["kwargs"]["target_jnt_range"]["r_elbow_flex"] == (0, 2.27)
    assert ids["myoFatiHandPoseRandom-v0"]["kwargs"]["muscle_condition"] == "fatigue"
    # ASL-derived target ranges (myobase/__init__.py:396-400)
    assert abs(registry.Rpos["mcp2_flexion"][1] - 1.30045) < 1e-9 and abs(registry.Rpos["ip_flexion"][0] + 1.309) < 1e-9


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors)."""
    from oracle.env_oracle import philox4x32_10
    c = philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in c] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    c = philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in c] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    c = philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in c] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_reach_oracle_obs_reward_match_reference_code(oracle_lib, models):
    """ReachEnvOracle's dict arithmetic vs vectors produced by the reference's reach_v0.py (ref_reach_env.npz)."""
    import collections
    from oracle.env_oracle import ReachEnvOracle
    g = np.load(os.path.join(G, "ref_reach_env.npz"))
    tips, tgts = list(g["tip_sids"]), list(g["target_sids"])
    env = ReachEnvOracle(models["hand"], tip_sids=tips, far_th=float(g["far_th"]))
    env.dt = float(g["dt"])
    for i in range(g["qpos"].shape[0]):
        site = g["site_xpos"][i]
        od = collections.OrderedDict()
        od["time"] = np.array([g["time"][i]]); od["qpos"] = g["qpos"][i]; od["qvel"] = g["qvel"][i] * env.dt
        od["act"] = g["act"][i]
        od["tip_pos"] = np.concatenate([site[s] for s in tips]); od["target_pos"] = np.concatenate([site[s] for s in tgts])
        od["reach_err"] = od["target_pos"] - od["tip_pos"]
        vec = np.concatenate([od[k].ravel() for k in ("qpos", "qvel", "tip_pos", "reach_err", "act")]).astype(np.float32)
        np.testing.assert_array_equal(vec, g["obs"][i])
        rd = env.get_reward_dict(od)
        for k in ("reach", "bonus", "act_reg", "penalty", "sparse", "solved", "done", "dense"):
            np.testing.assert_allclose(float(rd[k]), g[f"rwd_{k}"][i], rtol=1e-12, atol=1e-12, err_msg=k)


# ------------------------------------------------------------------------------------------------ libmujoco fixtures
def _mujoco_fixtures():
    import glob
    return sorted(glob.glob(os.path.join(G, "mujoco_*.npz")))


def fixture_model(g):
    """the compiled model a libmujoco fixture was written for: imported from the MJCF text the fixture carries (ids in MuJoCo's
    document order), or -- fixtures of multi-file models -- the synthetic model of that name, hash-checked"""
    from myosuite_amd.model import mjcf, synth
    if "xml" in g.files and str(g["xml"]):
        cm = mjcf.load(str(g["xml"])).compile()
    else:
        cm = synth.get_model(str(g["model"]))
    assert cm.hash() == str(g["model_hash"]), "fixture was written for another revision of the model / importer"
    return cm


def test_libmujoco_fixture_hook_is_wired():
    """tests/tools/validate_against_mujoco.py --write-fixture (run by anyone with `pip install mujoco`) drops
    tests/golden/mujoco_<model>.npz; this test and the two below it pick every such file up.  None is committed yet: the
    authoring image has no mujoco, which is why DESIGN.md says "engine parity unpinned"."""
    src = open(os.path.join(os.path.dirname(G), "tools", "validate_against_mujoco.py")).read()
    assert "--write-fixture" in src and "mujoco_" in src and "FORWARD_FIELDS" in src
    for f in _mujoco_fixtures():
        g = np.load(f)
        assert {"q0", "v0", "a0", "ctrl", "t_qpos", "model_hash", "f_qacc"} <= set(g.files), f


@pytest.mark.parametrize("path", _mujoco_fixtures() or [None])
def test_oracle_matches_libmujoco_fixture(oracle_lib, path):
    """fp64 oracle vs libmujoco (fixture written where mujoco is installed): compile-time constants, every stage of one
    mj_forward and the free-running trajectory.  Skipped while no fixture is committed."""
    if path is None:
        pytest.skip("no tests/golden/mujoco_*.npz committed (needs one run of tests/tools/validate_against_mujoco.py --write-fixture)")
    from myosuite_amd.model import synth
    from oracle import oracle as O
    g = np.load(path)
    cm = fixture_model(g)
    A = cm.arrays
    np.testing.assert_allclose(A["DOF_INVWEIGHT0"], g["c_dof_invweight0"], rtol=1e-5)
    np.testing.assert_allclose(A["ACT_ACC0"], g["c_actuator_acc0"], rtol=1e-5)
    np.testing.assert_allclose(A["ACT_LENGTHRANGE"].reshape(-1, 2), g["c_actuator_lengthrange"], rtol=1e-4, atol=1e-6)
    d = O.OracleData(O.OracleModel(cm))
    d.qpos[:] = g["q0"]; d.qvel[:] = g["v0"]; d.act[:] = g["a0"]; d.ctrl[:] = g["ctrl"][0]
    d.forward()
    for k in g.files:
        if k.startswith("f_") and k != "f_nefc":
            ref = g[k].ravel(); got = np.asarray(getattr(d, k[2:])).ravel()[:ref.size]
            if ref.size == 0:          # (a model without tendons / actuators: found by the fake-mujoco round trip)
                continue
            assert got.size == ref.size and np.abs(got - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k
    assert d.nefc == int(g["f_nefc"])

    def check_contacts(tag):
        """the colliders against libmujoco's contact list (fixtures written since round 5): multiplicity, order, distance, position, normal"""
        if f"k{tag}_ncon" not in g.files:
            return
        n = int(g[f"k{tag}_ncon"])
        assert d.ncon == n and d.nefc == int(g[f"k{tag}_nefc"]), (tag, d.ncon, n)
        if n:
            np.testing.assert_allclose(d.con_dist[:n], g[f"k{tag}_dist"], atol=2e-6)          # (the convex collider's own tolerance is 1e-6)
            np.testing.assert_allclose(d.con_pos[:n], g[f"k{tag}_pos"], atol=2e-6)
            np.testing.assert_allclose(d.con_frame[:n, :3], g[f"k{tag}_normal"], atol=1e-5)
    check_contacts(0)
    for s in range(g["ctrl"].shape[0]):
        d.ctrl[:] = g["ctrl"][s]; d.step()
        assert np.abs(d.qpos - g["t_qpos"][s]).max() < 1e-6 * max(1.0, np.abs(g["t_qpos"][s]).max()), s
        check_contacts(s + 1)


def test_libmujoco_validation_tool_round_trips_through_a_fake_mujoco(oracle_lib, tmp_path, monkeypatch, capsys):
    """tests/tools/validate_against_mujoco.py has never met a real `mujoco` module (none in this image).  So that its first real
    run cannot fail on plumbing, drive the whole tool -- MJCF dump, MjModel.from_xml_path, compile-time constants, the forward
    field mapping, the stepping loop, --write-fixture -- through tests/tools/fake_mujoco.py (the mujoco entry points it uses,
    implemented on the fp64 oracle), then feed the fixture it wrote to the reader the committed fixtures go through
    (test_oracle_matches_libmujoco_fixture).  The numbers are the oracle's own, so every difference must be ~0: what is tested
    is the writer / reader pair, the field names and shapes."""
    import importlib.util
    import json
    import sys
    tools = os.path.join(os.path.dirname(G), "tools")
    sys.path.insert(0, tools)
    try:
        import fake_mujoco
        fake_mujoco.install()
        spec = importlib.util.spec_from_file_location("validate_against_mujoco", os.path.join(tools, "validate_against_mujoco.py"))
        tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
        written = []
        real_savez = np.savez_compressed
        monkeypatch.setattr(np, "savez_compressed", lambda path, **kw: (written.append(str(tmp_path / os.path.basename(path))),
                                                                           real_savez(str(tmp_path / os.path.basename(path)), **kw))[1])
        for model, nsteps in (("hand", 12), ("contact_toy", 12), ("plane_toy", 90)):      # plane_toy: every primitive collider, bodies fall into contact
            monkeypatch.setattr(sys, "argv", ["validate_against_mujoco.py", "--model", model, "--steps", str(nsteps), "--write-fixture"])
            assert tool.main() == 0
            out = capsys.readouterr().out
            rep = json.loads(out[out.index("{"):out.rindex("}") + 1])
            assert rep["dims"]["nq"][0] == rep["dims"]["nq"][1] and rep["dims"]["ntendon"][0] == rep["dims"]["ntendon"][1]
            assert max(rep["compile_constants_maxabs_diff"].values()) < 1e-6, rep["compile_constants_maxabs_diff"]
            assert all(v < 1e-9 for k, v in rep["forward_rel_diff"].items() if k != "nefc") and rep["forward_rel_diff"]["nefc"][0] == rep["forward_rel_diff"]["nefc"][1]
            assert rep["qpos_divergence"]["oracle_vs_mujoco"]["max"] < 1e-9
        assert len(written) == 3 and all(os.path.exists(w) for w in written)
        gp = np.load(written[2])          # the contact lists travel with the fixture (start / middle / end of the trajectory)
        assert {"k0_ncon", "k45_ncon", "k90_ncon", "k90_dist", "k90_pos", "k90_normal", "k90_nefc"} <= set(gp.files)
        assert int(gp["k90_ncon"]) >= 3 and gp["k90_pos"].shape == (int(gp["k90_ncon"]), 3)
        g = np.load(written[0])
        assert str(g["mujoco_version"]) == "fake-oracle" and g["t_qpos"].shape[0] == 12
        assert {"q0", "v0", "a0", "ctrl", "t_qpos", "model_hash", "f_qacc", "f_nefc", "c_meaninertia"} <= set(g.files)
        for w in written:                                  # the reader of committed fixtures accepts what the writer wrote
            test_oracle_matches_libmujoco_fixture(oracle_lib, w)
    finally:
        sys.modules.pop("mujoco", None)
        sys.path.remove(tools)
