"""Generate golden vectors by EXECUTING the reference's own Python (importable parts only).

Run in the authoring container (needs /root/reference; the GPU box does not have it):
    python tests/golden/make_golden_reference.py
Writes tests/golden/ref_*.npz, which are committed.

What can be executed from /root/reference without `mujoco`/`gym` (both absent here):
  * myosuite/envs/myo/fatigue.py            CumulativeFatigue (3CC-r)        -> ref_fatigue.npz
  * myosuite/envs/myo/myobase/pose_v0.py    get_obs_dict / get_reward_dict    -> ref_pose_env.npz
  * myosuite/envs/obs_vec_dict.py           obsdict2obsvec                    -> ref_pose_env.npz
  * myosuite/envs/myo/myobase/reach_v0.py   get_obs_dict / get_reward_dict    -> ref_reach_env.npz
  * myosuite/envs/myo/myobase/walk_v0.py    get_obs_dict / get_reward_dict    -> ref_walk_env.npz
  * myosuite/envs/myo/myobase/walk_v0.py    ReachEnvV0 (leg stand) obs / reward / generate_qpos -> ref_stand_env.npz
  * myosuite/envs/myo/myobase/reorient_sar_v0.py  get_obs_dict / get_reward_dict -> ref_reorient_env.npz
  * myosuite/envs/myo/myobase/pen_v0.py     get_obs_dict / get_reward_dict    -> ref_pen_env.npz
  * myosuite/envs/myo/myobase/obj_hold_v0.py get_obs_dict / get_reward_dict   -> ref_objhold_env.npz
  * myosuite/envs/myo/myobase/key_turn_v0.py get_obs_dict / get_reward_dict   -> ref_keyturn_env.npz
  * myosuite/utils/quat_math.py, vector_math.py                               -> ref_math.npz
`mujoco` and `myosuite.utils.gym` are replaced by stubs that only provide the names those files
touch at import time (mjtDyn.mjDYN_MUSCLE, gym.utils.seeding.np_random, EzPickle); no arithmetic
is stubbed.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/myosuite"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path, stubs):
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def _stubs():
    mj = types.ModuleType("mujoco")
    mj.mjtDyn = types.SimpleNamespace(mjDYN_MUSCLE=4)
    gym = types.SimpleNamespace(
        utils=types.SimpleNamespace(
            seeding=types.SimpleNamespace(np_random=lambda seed=None: (np.random.default_rng(seed), seed)),
            EzPickle=type("EzPickle", (), {"__init__": lambda self, *a, **k: None})))
    utils = types.ModuleType("myosuite.utils")
    utils.gym = gym
    pkg = types.ModuleType("myosuite")
    base = types.ModuleType("myosuite.envs.myo.base_v0")
    base.BaseV0 = object
    return {"mujoco": mj, "myosuite": pkg, "myosuite.utils": utils, "myosuite.envs": types.ModuleType("myosuite.envs"),
            "myosuite.envs.myo": types.ModuleType("myosuite.envs.myo"), "myosuite.envs.myo.base_v0": base}


def gen_fatigue():
    fat = _load("ref_fatigue", f"{REF}/envs/myo/fatigue.py", _stubs())
    out = {}
    # (a) the reference's own test sequence (tests/mjx/test_fatigue.py:173-214): 5 muscles, frame_skip 5.
    #     myofinger's tau values live in the missing XML; MuJoCo muscle defaults (0.01, 0.04) are used.
    for tag, na, fs, tau in (("seq5", 5, 5, (0.01, 0.04)), ("rand39", 39, 10, (0.01, 0.04))):
        model = types.SimpleNamespace(
            opt=types.SimpleNamespace(timestep=0.002),
            actuator_dyntype=np.full(na, 4), actuator_dynprm=np.tile(np.array([tau[0], tau[1], 0.0]), (na, 1)))
        f = fat.CumulativeFatigue(model, frame_skip=fs, seed=0)
        if tag == "seq5":
            acts = [np.zeros(5), np.ones(5), np.array([0.3, 0.5, 0.7, 0.2, 0.8]), np.array([0.5] * 5)]
        else:
            rng = np.random.default_rng(7)
            acts = [rng.random(na) for _ in range(200)]
        MA, MR, MF = [], [], []
        for a in acts:
            ma, mr, mf = f.compute_act(np.asarray(a, dtype=np.float64))
            MA.append(ma.copy()); MR.append(mr.copy()); MF.append(mf.copy())
        out[f"{tag}_acts"] = np.array(acts); out[f"{tag}_MA"] = np.array(MA)
        out[f"{tag}_MR"] = np.array(MR); out[f"{tag}_MF"] = np.array(MF)
        out[f"{tag}_dt"] = np.array(0.002 * fs); out[f"{tag}_tau"] = np.array(tau)
    np.savez(os.path.join(OUT, "ref_fatigue.npz"), **out)


def gen_pose_env():
    pose = _load("ref_pose_v0", f"{REF}/envs/myo/myobase/pose_v0.py", _stubs())
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(11)
    out = {}
    for tag, nq, na, thd in (("elbow", 1, 6, 0.175), ("hand", 23, 39, 0.7)):
        n = 64
        qpos = rng.uniform(-1.5, 2.5, (n, nq)); qvel = rng.standard_normal((n, nq)) * 3
        act = rng.random((n, na)); target = rng.uniform(-1.0, 2.0, (n, nq))
        if tag == "elbow":          # include far / solved cases
            target[:8] = qpos[:8] + 7.0; target[8:16] = qpos[8:16] + 0.1
        else:
            target[:8] = qpos[:8] + 2.0; target[8:16] = qpos[8:16] + 0.05
        dt = 0.02
        obs, rwd = [], {k: [] for k in ("pose", "bonus", "penalty", "act_reg", "sparse", "solved", "done", "dense")}
        for i in range(n):
            self = types.SimpleNamespace(dt=dt, target_jnt_value=target[i], pose_thd=thd,
                                         mj_model=types.SimpleNamespace(na=na),
                                         rwd_keys_wt=pose.PoseEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS)
            data = types.SimpleNamespace(time=0.02 * i, qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy())
            od = pose.PoseEnvV0.get_obs_dict(self, self.mj_model, data)
            self.obs_dict = od
            ov = ovd.ObsVecDict()
            _, vec = ov.obsdict2obsvec(od, ["qpos", "qvel", "pose_err", "act"])     # key order: pose_v0.py:17 + base_v0.py:33-37
            # the reference expands dims to (1,1,d) before the reward (env_base.py:423-426)
            od3 = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
            self.obs_dict = od3
            rd = pose.PoseEnvV0.get_reward_dict(self, od3)
            obs.append(vec)
            for k in rwd:
                rwd[k].append(np.squeeze(rd[k]))
        out[f"{tag}_qpos"] = qpos; out[f"{tag}_qvel"] = qvel; out[f"{tag}_act"] = act; out[f"{tag}_target"] = target
        out[f"{tag}_obs"] = np.array(obs)
        for k in rwd:
            out[f"{tag}_rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
        out[f"{tag}_pose_thd"] = np.array(thd); out[f"{tag}_dt"] = np.array(dt)
    np.savez(os.path.join(OUT, "ref_pose_env.npz"), **out)


def gen_reach_env():
    """ReachEnvV0.get_obs_dict / get_reward_dict (reach_v0.py:95-151) executed on synthetic (qpos,qvel,act,site_xpos)."""
    reach = _load("ref_reach_v0", f"{REF}/envs/myo/myobase/reach_v0.py", _stubs())
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(5)
    n, nq, na, ntip = 48, 23, 39, 5
    tip_sids = [3, 7, 11, 15, 19]; target_sids = [20, 21, 22, 23, 24]
    qpos = rng.uniform(-1, 1.5, (n, nq)); qvel = rng.standard_normal((n, nq)) * 3; act = rng.random((n, na))
    site = rng.uniform(-0.3, 0.3, (n, 25, 3))
    site[:12, target_sids] = site[:12, tip_sids] + rng.uniform(-0.01, 0.01, (12, 5, 3))     # near: bonus / solved
    site[12:24, target_sids] = site[12:24, tip_sids] + 0.3                                  # far: penalty / done
    times = np.where(np.arange(n) % 3 == 0, 0.02, 0.5)                                      # far_th = inf while t <= 2 dt
    dt, far_th = 0.02, 0.034
    obs, rwd = [], {k: [] for k in ("reach", "bonus", "act_reg", "penalty", "sparse", "solved", "done", "dense")}
    for i in range(n):
        self = types.SimpleNamespace(dt=dt, far_th=far_th, tip_sids=tip_sids, target_sids=target_sids,
                                     mj_model=types.SimpleNamespace(na=na),
                                     rwd_keys_wt=reach.ReachEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS)
        data = types.SimpleNamespace(time=times[i], qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(),
                                     site_xpos=site[i].copy())
        od = reach.ReachEnvV0.get_obs_dict(self, self.mj_model, data)
        ov = ovd.ObsVecDict()
        _, vec = ov.obsdict2obsvec(od, ["qpos", "qvel", "tip_pos", "reach_err", "act"])
        od3 = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        self.obs_dict = od3
        rd = reach.ReachEnvV0.get_reward_dict(self, od3)
        obs.append(vec)
        for k in rwd:
            rwd[k].append(np.squeeze(rd[k]))
    out = dict(qpos=qpos, qvel=qvel, act=act, site_xpos=site, time=times, obs=np.array(obs), dt=np.array(dt),
               far_th=np.array(far_th), tip_sids=np.array(tip_sids), target_sids=np.array(target_sids))
    for k in rwd:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    np.savez(os.path.join(OUT, "ref_reach_env.npz"), **out)


def gen_walk_env():
    """WalkEnvV0.get_obs_dict / get_reward_dict (walk_v0.py:283-325 and the helpers :367-540) executed on synthetic
    mjData-like arrays with the synthetic leg's dimensions (15 bodies, nq 35, nv 34, 80 muscles)."""
    st = _stubs()
    st["myosuite.utils.quat_math"] = _load("ref_quat_math", f"{REF}/utils/quat_math.py", {})    # the reference's own module
    walk = _load("ref_walk_v0", f"{REF}/envs/myo/myobase/walk_v0.py", st)
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(9)
    n, nb, nq, nv, nu = 48, 15, 35, 34, 80
    body_ids = {"pelvis": 1, "torso": 2, "talus_r": 5, "talus_l": 11}
    jnt_ids = {"hip_flexion_r": 1, "hip_adduction_r": 2, "hip_rotation_r": 3, "hip_flexion_l": 15, "hip_adduction_l": 16,
               "hip_rotation_l": 17}
    jnt_qposadr = np.concatenate([[0], 7 + np.arange(28)])
    body_mass = np.concatenate([[0.0], rng.uniform(0.1, 30, nb - 1)])
    key0 = np.zeros(nq); key0[2] = 0.98; key0[3] = 0.7071067811865476; key0[6] = -0.7071067811865476
    qpos = key0 + rng.uniform(-0.3, 0.3, (n, nq)); qvel = rng.standard_normal((n, nv)) * 2
    quat = qpos[:, 3:7] + rng.uniform(-0.3, 0.3, (n, 4)); quat[n // 2:] = rng.standard_normal((n - n // 2, 4))
    qpos[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    act = rng.random((n, nu))
    xpos = rng.uniform(-1, 1, (n, nb, 3)); xpos[:, :, 2] = rng.uniform(0, 1.5, (n, nb))
    xipos = xpos + rng.uniform(-0.1, 0.1, (n, nb, 3)); xipos[: n // 3, :, 2] *= 0.5      # low COM -> done
    xquat = rng.standard_normal((n, nb, 4)); xquat /= np.linalg.norm(xquat, axis=2, keepdims=True)
    cvel = rng.standard_normal((n, nb, 6)); cvel[:, :, 4] -= 1.2
    alen = rng.uniform(0.05, 0.5, (n, nu)); avel = rng.standard_normal((n, nu)) * 60; afrc = -rng.uniform(0, 4000, (n, nu)) * 40
    steps = rng.integers(0, 400, n)
    dt = 0.01
    keys = list(walk.WalkEnvV0.DEFAULT_OBS_KEYS) + ["act"]
    obs = []; rk = ("vel_reward", "cyclic_hip", "ref_rot", "joint_angle_rew", "act_mag", "sparse", "solved", "done", "dense")
    rwd = {k: [] for k in rk}
    for i in range(n):
        model = types.SimpleNamespace(na=nu, body_mass=body_mass, jnt_qposadr=jnt_qposadr,
                                      body=lambda name: types.SimpleNamespace(id=body_ids[name]),
                                      joint=lambda name: types.SimpleNamespace(id=jnt_ids[name]))
        data = types.SimpleNamespace(time=dt * steps[i], qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(),
                                     xpos=xpos[i], xipos=xipos[i], xquat=xquat[i], cvel=cvel[i], actuator_length=alen[i],
                                     actuator_velocity=avel[i], actuator_force=afrc[i])
        env = object.__new__(walk.WalkEnvV0)
        env.mj_model = model; env.mj_data = data; env.dt = dt; env.steps = int(steps[i]); env.hip_period = 100
        env.target_x_vel = 0.0; env.target_y_vel = 1.2; env.target_rot = None; env.init_qpos = key0.copy()
        env.min_height = 0.8; env.max_rot = 0.8
        env.rwd_keys_wt = walk.WalkEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
        od = env.get_obs_dict(model, data)
        _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
        env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        rd = env.get_reward_dict(env.obs_dict)
        obs.append(vec)
        for k in rk:
            rwd[k].append(np.squeeze(rd[k]))
    out = dict(body_mass=body_mass, qpos=qpos, qvel=qvel, act=act, xpos=xpos, xipos=xipos, xquat=xquat, cvel=cvel,
               actuator_length=alen, actuator_velocity=avel, actuator_force=afrc, steps=steps, dt=np.array(dt),
               obs=np.array(obs), key0=key0,
               ids=np.array([body_ids["pelvis"], body_ids["torso"], body_ids["talus_l"], body_ids["talus_r"]] +
                            [int(jnt_qposadr[jnt_ids[k]]) for k in ("hip_flexion_l", "hip_flexion_r", "hip_adduction_l",
                                                                     "hip_adduction_r", "hip_rotation_l", "hip_rotation_r")]))
    for k in rk:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    np.savez(os.path.join(OUT, "ref_walk_env.npz"), **out)


def gen_stand_env():
    """walk_v0.ReachEnvV0 (myoLegStandRandom-v0): get_obs_dict / get_reward_dict (walk_v0.py:71-128) and generate_qpos
    (:153-168) executed on synthetic mjData-like arrays."""
    st = _stubs()
    st["myosuite.utils.quat_math"] = _load("ref_quat_math", f"{REF}/utils/quat_math.py", {})
    walk = _load("ref_walk_v0", f"{REF}/envs/myo/myobase/walk_v0.py", st)
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(19)
    n, nq, nv, nu, ns = 40, 35, 34, 80, 3
    tip, tgt = 1, 2
    qpos = rng.uniform(-1, 1, (n, nq)); qvel = rng.standard_normal((n, nv)) * rng.choice([0.02, 1.0], (n, 1)); act = rng.random((n, nu))
    site = rng.uniform(-0.5, 0.5, (n, ns, 3))
    dirs = rng.standard_normal((n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    site[:, tgt] = site[:, tip] + dirs * rng.uniform(0.0, 0.6, (n, 1))                 # around near_th .05 / .1 and far_th .44
    time = np.where(np.arange(n) % 4 == 0, 0.01, 0.5)                                  # time <= 2 dt: far_th = inf
    dt = 0.02
    keys = list(walk.ReachEnvV0.DEFAULT_OBS_KEYS) + ["act"]
    rk = ("reach", "bonus", "act_reg", "penalty", "sparse", "solved", "done", "dense")
    obs = []; rwd = {k: [] for k in rk}
    for i in range(n):
        model = types.SimpleNamespace(na=nu)
        data = types.SimpleNamespace(time=time[i], qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(), site_xpos=site[i])
        env = object.__new__(walk.ReachEnvV0)
        env.mj_model = model; env.dt = dt; env.tip_sids = [tip]; env.target_sids = [tgt]; env.far_th = 0.44
        env.rwd_keys_wt = walk.ReachEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
        od = env.get_obs_dict(model, data)
        _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
        env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        rd = env.get_reward_dict(env.obs_dict)
        obs.append(vec)
        for k in rk:
            rwd[k].append(np.squeeze(rd[k]))
    out = dict(qpos=qpos, qvel=qvel, act=act, tip=site[:, tip], target=site[:, tgt], time=time, dt=np.array(dt), obs=np.array(obs))
    for k in rk:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    # generate_qpos: init + U(range) on the jnt_qposadr entries, clipped to jnt_range (unlimited joints: range (0, 0))
    njnt = 29
    adr = np.concatenate([[0], 7 + np.arange(28)])
    jr = np.stack([-rng.uniform(0.1, 1.5, njnt), rng.uniform(0.1, 1.5, njnt)], 1); jr[[0, 5, 6]] = 0.0
    init = rng.uniform(-0.3, 0.3, nq)
    env = object.__new__(walk.ReachEnvV0)
    env.mj_model = types.SimpleNamespace(jnt_qposadr=adr, jnt_range=jr)
    env.init_qpos = init.copy(); env.joint_random_range = (-0.2, 0.2)
    env.np_random = np.random.default_rng(5)
    u = np.random.default_rng(5).uniform(low=-0.2, high=0.2, size=init.shape)          # the draw generate_qpos makes
    out.update(gq_adr=adr, gq_range=jr, gq_init=init, gq_draw=u, gq_out=env.generate_qpos())
    np.savez(os.path.join(OUT, "ref_stand_env.npz"), **out)


def gen_reorient_env():
    """ProprioceptiveEnvV0.get_obs_dict / get_reward_dict (reorient_sar_v0.py:116-174) executed on synthetic mjData-like
    arrays (nq 29, 39 muscles), plus the reference's euler2quat on the reset's desired-orientation draws."""
    st = _stubs()
    st["myosuite.utils.quat_math"] = _load("ref_quat_math", f"{REF}/utils/quat_math.py", {})
    st["myosuite.utils.vector_math"] = _load("ref_vector_math", f"{REF}/utils/vector_math.py", {})
    reo = _load("ref_reorient_sar_v0", f"{REF}/envs/myo/myobase/reorient_sar_v0.py", st)
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(21)
    n, nq, nu, nb, ng, ns = 48, 29, 39, 32, 6, 3
    obj_bid, eps_sid, t_gid, b_gid, tt_gid, tb_gid = 30, 1, 2, 3, 4, 5
    qpos = rng.uniform(-1, 1, (n, nq)); qvel = rng.standard_normal((n, nq)) * 3; act = rng.random((n, nu))
    xpos = rng.uniform(-0.5, 0.5, (n, nb, 3)); site = rng.uniform(-0.5, 0.5, (n, ns, 3))
    site[:16, eps_sid] = xpos[:16, obj_bid] + rng.uniform(-0.02, 0.02, (16, 3))       # near: not dropped
    gx = rng.uniform(-0.5, 0.5, (n, ng, 3))
    d1 = rng.standard_normal((n, 3)); d1 *= rng.uniform(0.03, 0.13, (n, 1)) / np.linalg.norm(d1, axis=1, keepdims=True)
    d2 = d1 + rng.standard_normal((n, 3)) * 0.02; d2[24:] = rng.standard_normal((n - 24, 3)) * 0.05
    gx[:, t_gid] = gx[:, b_gid] + d1; gx[:, tt_gid] = gx[:, tb_gid] + d2
    alen = rng.uniform(0.05, 0.4, (n, nu)); avel = rng.standard_normal((n, nu)); afrc = -rng.uniform(0, 300, (n, nu))
    dt, pen_length, tar_length = 0.01, 0.07, 0.07
    keys = ["hand_jnt", "obj_pos", "obj_vel", "obj_rot", "obj_des_rot", "obj_err_pos", "obj_err_rot", "mlen", "mvel", "mforce", "act"]
    rk = ("pos_align", "rot_align", "act_reg", "drop", "bonus", "sparse", "solved", "done", "dense")
    obs = []; rwd = {k: [] for k in rk}
    for i in range(n):
        model = types.SimpleNamespace(na=nu, site_rgba=np.zeros((ns, 4)))
        data = types.SimpleNamespace(time=0.1 * i, qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(), xpos=xpos[i],
                                     site_xpos=site[i], geom_xpos=gx[i], actuator_length=alen[i], actuator_velocity=avel[i],
                                     actuator_force=afrc[i])
        env = object.__new__(reo.ProprioceptiveEnvV0)
        env.mj_model = model; env.dt = dt; env.obj_bid = obj_bid; env.eps_ball_sid = eps_sid; env.obj_t_gid = t_gid
        env.obj_b_gid = b_gid; env.tar_t_gid = tt_gid; env.tar_b_gid = tb_gid; env.pen_length = pen_length
        env.tar_length = tar_length; env.success_indicator_sid = 2
        env.rwd_keys_wt = reo.ProprioceptiveEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
        od = env.get_obs_dict(model, data)
        _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
        env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        rd = env.get_reward_dict(env.obs_dict)
        obs.append(vec)
        for k in rk:
            rwd[k].append(np.squeeze(rd[k]))
    eul = np.stack([rng.uniform(-1, 1, 32), rng.uniform(-0.8, 1.2, 32), np.zeros(32)], 1)
    qm = st["myosuite.utils.quat_math"]
    out = dict(qpos=qpos, qvel=qvel, act=act, obj_xpos=xpos[:, obj_bid], eps_pos=site[:, eps_sid], top_minus_bot=d1,
               ttop_minus_tbot=d2, actuator_length=alen, actuator_velocity=avel, actuator_force=afrc, dt=np.array(dt),
               pen_length=np.array(pen_length), tar_length=np.array(tar_length), obs=np.array(obs), euler=eul,
               euler2quat=np.array([qm.euler2quat(e) for e in eul]), quat2mat=np.array([qm.quat2mat(qm.euler2quat(e)) for e in eul]))
    for k in rk:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    np.savez(os.path.join(OUT, "ref_reorient_env.npz"), **out)


def gen_pen_env():
    """PenTwirlFixedEnvV0.get_obs_dict / get_reward_dict (pen_v0.py:116-169) on synthetic mjData-like arrays."""
    st = _stubs()
    st["myosuite.utils.quat_math"] = _load("ref_quat_math", f"{REF}/utils/quat_math.py", {})
    st["myosuite.utils.vector_math"] = _load("ref_vector_math", f"{REF}/utils/vector_math.py", {})
    pen = _load("ref_pen_v0", f"{REF}/envs/myo/myobase/pen_v0.py", st)
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(31)
    n, nq, nu, nb, ns = 40, 29, 39, 32, 6
    obj_bid, eps_sid, ot, ob_, tt, tb = 30, 0, 1, 2, 3, 4
    qpos = rng.uniform(-1, 1, (n, nq)); qvel = rng.standard_normal((n, nq)) * 3; act = rng.random((n, nu))
    xpos = rng.uniform(-0.5, 0.5, (n, nb, 3)); site = rng.uniform(-0.5, 0.5, (n, ns, 3))
    site[:14, eps_sid] = xpos[:14, obj_bid] + rng.uniform(-0.02, 0.02, (14, 3))
    d1 = rng.standard_normal((n, 3)); d1 *= 0.13 / np.linalg.norm(d1, axis=1, keepdims=True)
    d2 = d1 + rng.standard_normal((n, 3)) * 0.03; d2[20:] = rng.standard_normal((n - 20, 3)) * 0.1
    site[:, ot] = site[:, ob_] + d1; site[:, tt] = site[:, tb] + d2
    dt = 0.01
    keys = list(pen.PenTwirlFixedEnvV0.DEFAULT_OBS_KEYS) + ["act"]
    rk = ("pos_align", "rot_align", "act_reg", "drop", "bonus", "sparse", "solved", "done", "dense")
    obs = []; rwd = {k: [] for k in rk}
    for i in range(n):
        model = types.SimpleNamespace(na=nu)
        data = types.SimpleNamespace(time=0.1 * i, qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(), xpos=xpos[i],
                                     site_xpos=site[i])
        env = object.__new__(pen.PenTwirlFixedEnvV0)
        env.mj_model = model; env.dt = dt; env.obj_bid = obj_bid; env.eps_ball_sid = eps_sid; env.obj_t_sid = ot
        env.obj_b_sid = ob_; env.tar_t_sid = tt; env.tar_b_sid = tb; env.pen_length = 0.13; env.tar_length = 0.13
        env.rwd_keys_wt = pen.PenTwirlFixedEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
        od = env.get_obs_dict(model, data)
        _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
        env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        rd = env.get_reward_dict(env.obs_dict)
        obs.append(vec)
        for k in rk:
            rwd[k].append(np.squeeze(rd[k]))
    out = dict(qpos=qpos, qvel=qvel, act=act, obj_xpos=xpos[:, obj_bid], eps_pos=site[:, eps_sid], top_minus_bot=d1,
               ttop_minus_tbot=d2, dt=np.array(dt), obs=np.array(obs), keys=np.array(keys))
    for k in rk:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    np.savez(os.path.join(OUT, "ref_pen_env.npz"), **out)


def gen_objhold_env():
    """ObjHoldFixedEnvV0.get_obs_dict / get_reward_dict (obj_hold_v0.py:82-131) on synthetic mjData-like arrays."""
    hold = _load("ref_obj_hold_v0", f"{REF}/envs/myo/myobase/obj_hold_v0.py", _stubs())
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(41)
    n, nq, nv, nu, ns = 40, 30, 29, 39, 4
    obj_sid, goal_sid = 2, 0
    qpos = rng.uniform(-1, 1, (n, nq)); qvel = rng.standard_normal((n, nv)) * 3; act = rng.random((n, nu))
    site = rng.uniform(-0.5, 0.5, (n, ns, 3))
    site[:12, goal_sid] = site[:12, obj_sid] + rng.uniform(-0.008, 0.008, (12, 3))      # near: bonus / solved
    site[12:20, goal_sid] = site[12:20, obj_sid] + 0.25                                  # far: drop / done
    dt = 0.02
    keys = list(hold.ObjHoldFixedEnvV0.DEFAULT_OBS_KEYS) + ["act"]
    rk = ("goal_dist", "bonus", "act_reg", "penalty", "sparse", "solved", "done", "dense")
    obs = []; rwd = {k: [] for k in rk}
    for i in range(n):
        model = types.SimpleNamespace(na=nu)
        data = types.SimpleNamespace(time=0.1 * i, qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(), site_xpos=site[i])
        env = object.__new__(hold.ObjHoldFixedEnvV0)
        env.mj_model = model; env.dt = dt; env.object_sid = obj_sid; env.goal_sid = goal_sid
        env.rwd_keys_wt = hold.ObjHoldFixedEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
        od = env.get_obs_dict(model, data)
        _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
        env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
        rd = env.get_reward_dict(env.obs_dict)
        obs.append(vec)
        for k in rk:
            rwd[k].append(np.squeeze(rd[k]))
    out = dict(qpos=qpos, qvel=qvel, act=act, obj_pos=site[:, obj_sid], goal_pos=site[:, goal_sid], dt=np.array(dt), obs=np.array(obs))
    for k in rk:
        out[f"rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    np.savez(os.path.join(OUT, "ref_objhold_env.npz"), **out)


def gen_keyturn_env():
    """KeyTurnEnvV0.get_obs_dict / get_reward_dict (key_turn_v0.py:101-150) on synthetic mjData-like arrays."""
    kt = _load("ref_key_turn_v0", f"{REF}/envs/myo/myobase/key_turn_v0.py", _stubs())
    ovd = _load("ref_obs_vec_dict", f"{REF}/envs/obs_vec_dict.py", {})
    rng = np.random.default_rng(43)
    n, nq, nu, ns = 48, 24, 39, 5
    kh, IF, TH = 4, 1, 0
    qpos = rng.uniform(-1, 1, (n, nq)); qvel = rng.standard_normal((n, nq)) * 3; act = rng.random((n, nu))
    qpos[:, -1] = rng.uniform(-2.0, 7.0, n)                                            # key angle: below / above pi/2, pi, 2 pi
    site = rng.uniform(-0.5, 0.5, (n, ns, 3))
    dirs = rng.standard_normal((n, 2, 3)); dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    rad = rng.uniform(0.0, 0.16, (n, 2))                                               # around the 0.03 shell, the 0.05 / 0.1 bounds
    site[:, IF] = site[:, kh] - dirs[:, 0] * rad[:, :1]; site[:, TH] = site[:, kh] - dirs[:, 1] * rad[:, 1:]
    dt = 0.02
    keys = list(kt.KeyTurnEnvV0.DEFAULT_OBS_KEYS) + ["act"]
    rk = ("key_turn", "IFtip_approach", "THtip_approach", "act_reg", "bonus", "penalty", "sparse", "solved", "done", "dense")
    out = {}
    for goal_th in (3.14, 2 * np.pi):
        obs = []; rwd = {k: [] for k in rk}
        for i in range(n):
            model = types.SimpleNamespace(na=nu)
            data = types.SimpleNamespace(time=0.1 * i, qpos=qpos[i].copy(), qvel=qvel[i].copy(), act=act[i].copy(), site_xpos=site[i])
            env = object.__new__(kt.KeyTurnEnvV0)
            env.mj_model = model; env.dt = dt; env.keyhead_sid = kh; env.IF_sid = IF; env.TH_sid = TH; env.goal_th = goal_th
            env.rwd_keys_wt = kt.KeyTurnEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS
            od = env.get_obs_dict(model, data)
            _, vec = ovd.ObsVecDict().obsdict2obsvec(od, keys)
            env.obs_dict = {k: np.asarray(v)[None, None, :] for k, v in od.items()}
            rd = env.get_reward_dict(env.obs_dict)
            obs.append(vec)
            for k in rk:
                rwd[k].append(np.squeeze(rd[k]))
        tag = "a" if goal_th == 3.14 else "b"
        out[f"{tag}_goal_th"] = np.array(goal_th); out[f"{tag}_obs"] = np.array(obs)
        for k in rk:
            out[f"{tag}_rwd_{k}"] = np.array(rwd[k], dtype=np.float64)
    out.update(qpos=qpos, qvel=qvel, act=act, keyhead=site[:, kh], iftip=site[:, IF], thtip=site[:, TH], dt=np.array(dt))
    np.savez(os.path.join(OUT, "ref_keyturn_env.npz"), **out)


def gen_math():
    qm = _load("ref_quat_math", f"{REF}/utils/quat_math.py", {})
    vm = _load("ref_vector_math", f"{REF}/utils/vector_math.py", {})
    rng = np.random.default_rng(3)
    eul = rng.uniform(-1.2, 1.2, (32, 3))
    quat = np.array([qm.euler2quat(e) for e in eul])
    mat = np.array([qm.quat2mat(q) for q in quat])
    v1 = rng.standard_normal((32, 3)); v2 = rng.standard_normal((32, 3))
    cosv = vm.calculate_cosine(v1, v2)
    qa = rng.standard_normal((32, 4)); qa /= np.linalg.norm(qa, axis=1, keepdims=True)
    qb = rng.standard_normal((32, 4)); qb /= np.linalg.norm(qb, axis=1, keepdims=True)
    mul = np.array([qm.mulQuat(a, b) for a, b in zip(qa, qb)])
    np.savez(os.path.join(OUT, "ref_math.npz"), euler=eul, euler2quat=quat, quat2mat=mat, v1=v1, v2=v2,
             cosine=cosv, qa=qa, qb=qb, mulQuat=mul)


if __name__ == "__main__":
    gen_fatigue()
    gen_pose_env()
    gen_reach_env()
    gen_walk_env()
    gen_stand_env()
    gen_reorient_env()
    gen_pen_env()
    gen_objhold_env()
    gen_keyturn_env()
    gen_math()
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("ref_")))
