"""numpy/fp64 restatement of the reference's env-level arithmetic around the physics step.

TEST INFRASTRUCTURE ONLY (see oracle/mmo_engine.c header): used by tests/, smoke() and
bench.py's cpu_baseline leg.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import collections

import numpy as np

from . import oracle as O


# ---------------------------------------------------------------------- Philox4x32-10
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011); mirrors csrc/myosim_engine.hip:philox4x32_10."""
    c = [np.asarray(x, dtype=np.uint64) for x in (c0, c1, c2, c3)]
    k0 = np.uint64(k0); k1 = np.uint64(k1)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask
        n3 = p0 & mask
        c = [n0, n1, n2, n3]
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return c


def u01(x):
    return ((np.asarray(x, np.uint64) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def uniform_stream(n: int, seed: int, stream_id: int) -> np.ndarray:
    """Same values as mm_uniform(out, n, seed, stream_id)."""
    n4 = (n + 3) // 4
    i4 = np.arange(n4, dtype=np.uint64)
    c = philox4x32_10(i4 & np.uint64(0xFFFFFFFF), i4 >> np.uint64(32), np.uint64(stream_id & 0xFFFFFFFF),
                      np.uint64(stream_id >> 32), seed & 0xFFFFFFFF, seed >> 32)
    out = np.stack([u01(x) for x in c], axis=1).reshape(-1)
    return out[:n]


def pose_reset_draws(nq: int, env: int, episode: int, seed: int):
    """(u_qpos[nq], u_target[nq]) exactly as k_reset draws them."""
    i = np.arange(nq)
    c = philox4x32_10((i >> 1).astype(np.uint64), np.zeros(nq, np.uint64), np.full(nq, env, np.uint64),
                      np.full(nq, episode, np.uint64), seed & 0xFFFFFFFF, seed >> 32)
    u = np.stack([u01(x) for x in c], axis=1)   # [nq, 4]
    return u[i, i & 1], u[i, 2 + (i & 1)]


def env_draw(ncomp: int, env: int, episode: int, seed: int, stream_id: int):
    """u[ncomp] exactly as mm_env_draw draws a per-env model delta (counter = (k/4, stream_id, env, episode), word k%4)."""
    k = np.arange(ncomp)
    c = philox4x32_10((k >> 2).astype(np.uint64), np.full(ncomp, stream_id, np.uint64), np.full(ncomp, env, np.uint64),
                      np.full(ncomp, episode, np.uint64), seed & 0xFFFFFFFF, seed >> 32)
    u = np.stack([u01(x) for x in c], axis=1)
    return u[k, k & 3]


# ---------------------------------------------------------------------- fatigue (3CC-r)
class FatigueOracle:
    """myosuite/envs/myo/fatigue.py:6-99, restated (no mujoco dependency: tau values are passed in)."""

    def __init__(self, tauact, taudeact, dt, na):
        self._r = 10 * 15; self._F = 0.00912; self._R = 0.1 * 0.00094      # fatigue.py:9-11
        self._dt = dt
        self.na = na
        self._tauact = np.asarray(tauact, np.float64); self._taudeact = np.asarray(taudeact, np.float64)
        self.reset()

    def reset(self, fatigue_reset_vec=None):
        if fatigue_reset_vec is not None:                                      # fatigue.py:91-95
            self._MF = np.asarray(fatigue_reset_vec, np.float64).copy()
            self._MR = 1 - self._MF
            self._MA = np.zeros(self.na)
        else:                                                                  # fatigue.py:96-99
            self._MA = np.zeros(self.na); self._MR = np.ones(self.na); self._MF = np.zeros(self.na)

    def compute_act(self, act):                                                # fatigue.py:38-76
        TL = np.asarray(act, np.float64).copy()
        MA, MR, MF = self._MA, self._MR, self._MF
        LD = 1 / self._tauact * (0.5 + 1.5 * MA)
        LR = (0.5 + 1.5 * MA) / self._taudeact
        C = np.zeros_like(MA)
        i1 = (MA < TL) & (MR > (TL - MA)); C[i1] = LD[i1] * (TL[i1] - MA[i1])
        i2 = (MA < TL) & (MR <= (TL - MA)); C[i2] = LD[i2] * MR[i2]
        i3 = MA >= TL; C[i3] = LR[i3] * (TL[i3] - MA[i3])
        rR = np.where(MA >= TL, self._r * self._R, self._R)
        C = np.clip(C, np.maximum(-MA / self._dt + self._F * MA, (MR - 1) / self._dt + rR * MF),
                    np.minimum((1 - MA) / self._dt + self._F * MA, MR / self._dt + rR * MF))
        dMA = (C - self._F * MA) * self._dt
        dMR = (-C + rR * MF) * self._dt
        dMF = (self._F * MA - rR * MF) * self._dt
        self._MA = MA + dMA; self._MR = MR + dMR; self._MF = MF + dMF
        return self._MA, self._MR, self._MF


# ---------------------------------------------------------------------- Pose env (single env)
class PoseEnvOracle:
    """Single-env CPU restatement of PoseEnvV0 on the fp64 oracle engine.

    step():   BaseV0.step (myosuite/envs/myo/base_v0.py:82-118) -> Robot.step (robot/robot.py:864-933,
              n_frames x mj_step) -> MujocoEnv.forward (envs/env_base.py:409-432): mj_forward on the new
              state (robot.py:595-607), get_obs_dict / get_reward_dict (myobase/pose_v0.py:100-140).
    """
    RWD_KEYS_WT = {"pose": 1.0, "bonus": 4.0, "act_reg": 1.0, "penalty": 50}

    def __init__(self, compiled, pose_thd, frame_skip=10, normalize_act=True, muscle_condition="",
                 weighted_reward_keys=None, reaf=None):
        self.cm = compiled
        self.om = O.OracleModel(compiled)
        self.d = O.OracleData(self.om)
        self.pose_thd = pose_thd
        self.frame_skip = frame_skip
        self.normalize_act = normalize_act
        self.muscle_condition = muscle_condition
        self.rwd_keys_wt = dict(weighted_reward_keys or self.RWD_KEYS_WT)
        self.dt = compiled.timestep * frame_skip
        self.target_jnt_value = np.zeros(compiled.nq)
        self.reaf = reaf
        self.far_th_pose = 4 * np.pi / 2                         # pose_v0.py:118 (torso_v0.py:119 uses pi)
        self.muscle = compiled.arrays["ACT_DYNTYPE"] == 4
        if muscle_condition == "fatigue":
            dyn = compiled.arrays["ACT_DYNPRM"].reshape(-1, 3).astype(np.float64)
            self.fatigue = FatigueOracle(dyn[self.muscle, 0], dyn[self.muscle, 1], self.dt, int(self.muscle.sum()))
        self.steps = 0

    def reset(self, qpos, target, qvel=None):
        self.d.reset()                                           # mj_resetData (robot.py:999)
        self.d.qpos[:] = qpos
        if qvel is not None:
            self.d.qvel[:] = qvel
        self.target_jnt_value = np.asarray(target, np.float64).copy()
        self.steps = 0
        if self.muscle_condition == "fatigue":
            self.fatigue.reset()
        return self.get_obs()

    def get_obs_dict(self):                                      # pose_v0.py:100-111
        d = self.d
        od = collections.OrderedDict()
        od["time"] = np.array([d.time])
        od["qpos"] = d.qpos.copy()
        od["qvel"] = d.qvel.copy() * self.dt
        od["act"] = d.act.copy() if self.cm.na > 0 else np.zeros_like(od["qpos"])
        od["pose_err"] = self.target_jnt_value - od["qpos"]
        return od

    def get_obs(self):                                           # obs_vec_dict.py:76-88, keys pose_v0.py:17 + act
        od = self.get_obs_dict()
        self.obs_dict = od
        keys = ("qpos", "qvel", "pose_err") + (("act",) if self.cm.na > 0 else ())        # base_v0.py:33-37
        return np.concatenate([od[k].ravel() for k in keys]).astype(np.float32)

    def get_reward_dict(self, od):                               # pose_v0.py:113-140
        pose_dist = np.linalg.norm(od["pose_err"], axis=-1)
        act_mag = np.linalg.norm(od["act"], axis=-1)
        if self.cm.na != 0:
            act_mag = act_mag / self.cm.na
        far_th = self.far_th_pose
        rwd = collections.OrderedDict((
            ("pose", -1.0 * pose_dist),
            ("bonus", 1.0 * (pose_dist < self.pose_thd) + 1.0 * (pose_dist < 1.5 * self.pose_thd)),
            ("penalty", -1.0 * (pose_dist > far_th)),
            ("act_reg", -1.0 * act_mag),
            ("sparse", -1.0 * pose_dist),
            ("solved", pose_dist < self.pose_thd),
            ("done", pose_dist > far_th)))
        rwd["dense"] = np.sum([wt * rwd[k] for k, wt in self.rwd_keys_wt.items()], axis=0)
        return rwd

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:                    # base_v0.py:86-90
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        elif self.normalize_act:                                 # base_v0.py:94-96 -> robot.py:786-796
            cr = self.cm.arrays["ACT_CTRLRANGE"].reshape(-1, 2)
            ctrl = cr.mean(axis=-1) + ctrl * (cr[:, 1] - cr[:, 0]) / 2.0
        if self.muscle_condition == "fatigue":                   # base_v0.py:99-103
            ctrl[self.muscle], _, _ = self.fatigue.compute_act(ctrl[self.muscle])
        elif self.muscle_condition == "reafferentation":         # base_v0.py:104-108
            src, dst = self.reaf
            ctrl[dst] = ctrl[src]
            ctrl[src] = 0
        self.d.ctrl[:] = ctrl                                    # robot.py:902
        self.last_ctrl = ctrl
        self.d.step(self.frame_skip)                             # robot.py:856-861
        self.d.forward()                                         # robot.py:607 (sensor2sim)
        self.steps += 1
        obs = self.get_obs()
        rwd = self.get_reward_dict(self.obs_dict)
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd


def reach_reset_draws(n3: int, env: int, episode: int, seed: int):
    """u[n3] exactly as k_reset draws the reach targets (counter = (i/4, 1, env, episode), word i%4)."""
    i = np.arange(n3)
    c = philox4x32_10((i >> 2).astype(np.uint64), np.ones(n3, np.uint64), np.full(n3, env, np.uint64),
                      np.full(n3, episode, np.uint64), seed & 0xFFFFFFFF, seed >> 32)
    u = np.stack([u01(x) for x in c], axis=1)
    return u[i, i & 3]


class ReachEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of ReachEnvV0 (myosuite/envs/myo/myobase/reach_v0.py:95-151)."""
    RWD_KEYS_WT = {"reach": 1.0, "bonus": 4.0, "penalty": 50}

    def __init__(self, compiled, tip_sids, far_th, frame_skip=10, normalize_act=True, muscle_condition="", reaf=None):
        super().__init__(compiled, pose_thd=0.0, frame_skip=frame_skip, normalize_act=normalize_act,
                         muscle_condition=muscle_condition, weighted_reward_keys=self.RWD_KEYS_WT, reaf=reaf)
        self.tip_sids = list(tip_sids)
        self.far_th = far_th
        self.target_pos = np.zeros(3 * len(self.tip_sids))

    def reset(self, target_pos):
        self.d.reset()
        self.target_pos = np.asarray(target_pos, np.float64).copy()
        self.steps = 0
        if self.muscle_condition == "fatigue":
            self.fatigue.reset()
        self.d.forward()
        return self.get_obs()

    def get_obs_dict(self):                                      # reach_v0.py:95-121
        d = self.d
        od = collections.OrderedDict()
        od["time"] = np.array([d.time])
        od["qpos"] = d.qpos.copy()
        od["qvel"] = d.qvel.copy() * self.dt
        od["act"] = d.act.copy()
        od["tip_pos"] = np.concatenate([d.site_xpos[s] for s in self.tip_sids])
        od["target_pos"] = self.target_pos.copy()
        od["reach_err"] = od["target_pos"] - od["tip_pos"]
        return od

    def get_obs(self):
        od = self.get_obs_dict()
        self.obs_dict = od
        keys = ("qpos", "qvel", "tip_pos", "reach_err") + (("act",) if self.cm.na > 0 else ())
        return np.concatenate([od[k].ravel() for k in keys]).astype(np.float32)

    def get_reward_dict(self, od):                               # reach_v0.py:123-151
        reach_dist = np.linalg.norm(od["reach_err"], axis=-1)
        act_mag = np.linalg.norm(od["act"], axis=-1) / self.cm.na if self.cm.na != 0 else 0
        far_th = self.far_th * len(self.tip_sids) if np.squeeze(od["time"]) > 2 * self.dt else np.inf
        near_th = len(self.tip_sids) * 0.0125
        rwd = collections.OrderedDict((
            ("reach", -1.0 * reach_dist),
            ("bonus", 1.0 * (reach_dist < 2 * near_th) + 1.0 * (reach_dist < near_th)),
            ("act_reg", -1.0 * act_mag),
            ("penalty", -1.0 * (reach_dist > far_th)),
            ("sparse", -1.0 * reach_dist),
            ("solved", reach_dist < near_th),
            ("done", reach_dist > far_th)))
        rwd["dense"] = np.sum([wt * rwd[k] for k, wt in self.rwd_keys_wt.items()], axis=0)
        return rwd


# ---------------------------------------------------------------------- WalkEnvV0
def walk_reset_draws(nq: int, env: int, episode: int, seed: int):
    """(coin, noise[nq]) of the device-side walk reset (csrc/myosim_engine.hip:k_reset, walk branch):
    coin = u01(philox(0xFFFF, 2, env, episode)[0]); noise_i = 0.02 * sqrt(-2 ln u1) cos(2 pi u2), float32."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = philox4x32_10(0xFFFF, 2, env, episode, k0, k1)
    coin = float(u01(c[0]))
    i = np.arange(nq)
    c = philox4x32_10(i, np.full(nq, 2), np.full(nq, env), np.full(nq, episode), k0, k1)
    u1 = (((c[0] >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)).astype(np.float32)
    u2 = u01(c[1])
    z = np.float32(0.02) * np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.283185307179586) * u2)
    z = z.astype(np.float32)
    z[2:7] = 0
    return coin, z


def walk_obs_reward(body_mass, qpos, qvel, act, xpos, xipos, xquat, cvel, actuator_length, actuator_velocity,
                    actuator_force, steps, dt, ids, prm, rwd_keys_wt):
    """WalkEnvV0.get_obs_dict + obsdict2obsvec + get_reward_dict on raw arrays (walk_v0.py:283-325, 367-540).
    ids: dict(pelvis, torso, talus_l, talus_r body ids; qadr of hip_flexion_l/r, hip_adduction_l/r, hip_rotation_l/r);
    prm: dict(min_height, max_rot, hip_period, target_x_vel, target_y_vel, target_rot)."""
    mass = np.asarray(body_mass, np.float64)[:, None]
    com_vel = (np.sum(mass * -cvel, 0) / np.sum(mass))[3:5]                       # walk_v0.py:438-446
    com = np.sum(mass * xipos, 0) / np.sum(mass)                                  # walk_v0.py:528-535
    height = com[2]
    feet_heights = np.array([xpos[ids["talus_l"]][2], xpos[ids["talus_r"]][2]])   # walk_v0.py:400-412
    feet_rel = np.array([xpos[ids["talus_l"]] - xpos[ids["pelvis"]], xpos[ids["talus_r"]] - xpos[ids["pelvis"]]])
    phase = (steps / prm["hip_period"]) % 1                                       # walk_v0.py:291
    od = collections.OrderedDict(
        qpos_without_xy=qpos[2:].copy(), qvel=qvel * dt, com_vel=com_vel, torso_angle=xquat[ids["torso"]].copy(),
        feet_heights=feet_heights, height=np.array([height]), feet_rel_positions=feet_rel.ravel(),
        phase_var=np.array([phase]), muscle_length=actuator_length.copy(),
        muscle_velocity=np.clip(actuator_velocity, -100, 100), muscle_force=np.clip(actuator_force / 1000, -100, 100),
        act=act.copy())
    obs = np.concatenate([np.asarray(v, np.float64).ravel() for v in od.values()])
    vel_reward = np.exp(-np.square(prm["target_y_vel"] - com_vel[1])) + np.exp(-np.square(prm["target_x_vel"] - com_vel[0]))
    des = np.array([0.8 * np.cos(phase * 2 * np.pi + np.pi), 0.8 * np.cos(phase * 2 * np.pi)], dtype=np.float32)
    ang = np.array([qpos[ids["hip_flexion_l"]], qpos[ids["hip_flexion_r"]]])
    cyclic_hip = np.linalg.norm(des - ang)                                        # walk_v0.py:453-468
    ref_rot = np.exp(-np.linalg.norm(5.0 * (qpos[3:7] - np.asarray(prm["target_rot"], np.float64))))
    ja = np.array([qpos[ids[k]] for k in ("hip_adduction_l", "hip_adduction_r", "hip_rotation_l", "hip_rotation_r")])
    joint_angle_rew = np.exp(-5 * np.mean(np.abs(ja)))                            # walk_v0.py:390-398
    na = act.size
    act_mag = np.linalg.norm(act) / na if na else 0.0
    q = qpos[3:7]; nq_ = np.sum(q * q)
    r00 = 1.0 - (2.0 / nq_) * (q[2] * q[2] + q[3] * q[3])                         # quat_math.py:151-174
    done = 1 if (height < prm["min_height"] or abs(r00) > prm["max_rot"]) else 0   # walk_v0.py:382-388,514-526
    rwd = collections.OrderedDict((
        ("vel_reward", vel_reward), ("cyclic_hip", cyclic_hip), ("ref_rot", ref_rot), ("joint_angle_rew", joint_angle_rew),
        ("act_mag", act_mag), ("sparse", vel_reward), ("solved", vel_reward >= 1.0), ("done", done)))
    rwd["dense"] = np.sum([wt * rwd[k] for k, wt in rwd_keys_wt.items()], axis=0)
    return obs, rwd


class WalkEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of WalkEnvV0 (walk_v0.py:189-540) on the fp64 oracle engine."""
    RWD_KEYS_WT = {"vel_reward": 5.0, "done": -100, "cyclic_hip": -10, "ref_rot": 10.0, "joint_angle_rew": 5.0}

    def __init__(self, compiled, frame_skip=10, normalize_act=True, muscle_condition="", min_height=0.8, max_rot=0.8,
                 hip_period=100, target_x_vel=0.0, target_y_vel=1.2, target_rot=None):
        super().__init__(compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        cm = compiled
        qadr = cm.arrays["JNT_QPOSADR"]
        self.ids = {b: cm.body_id(b) for b in ("pelvis", "torso", "talus_l", "talus_r")}
        for j in ("hip_flexion_l", "hip_flexion_r", "hip_adduction_l", "hip_adduction_r", "hip_rotation_l", "hip_rotation_r"):
            self.ids[j] = int(qadr[cm.joint_id(j)])
        rot = target_rot if target_rot is not None else cm.key_qpos[0][3:7].astype(np.float32)
        self.prm = dict(min_height=min_height, max_rot=max_rot, hip_period=hip_period, target_x_vel=target_x_vel,
                        target_y_vel=target_y_vel, target_rot=np.asarray(rot, np.float64))

    def reset(self, qpos, qvel):
        self.d.reset()
        self.d.qpos[:] = qpos; self.d.qvel[:] = qvel
        self.steps = 0
        if self.muscle_condition == "fatigue":
            self.fatigue.reset()
        self.d.ctrl[:] = 0
        self.d.forward()
        return self._obs_rwd()[0]

    def _obs_rwd(self):
        d = self.d
        return walk_obs_reward(self.cm.arrays["BODY_MASS"], d.qpos, d.qvel, d.act, d.xpos, d.xipos, d.xquat, d.cvel,
                               d.actuator_length, d.actuator_velocity, d.actuator_force, self.steps, self.dt, self.ids,
                               self.prm, self.rwd_keys_wt)

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        if self.muscle_condition == "fatigue":
            ctrl[self.muscle], _, _ = self.fatigue.compute_act(ctrl[self.muscle])
        self.d.ctrl[:] = ctrl
        self.last_ctrl = ctrl
        self.d.step(self.frame_skip)
        self.d.forward()
        obs, rwd = self._obs_rwd()            # self.steps is incremented AFTER the base step (walk_v0.py:354-357)
        self.steps += 1
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd


# ---------------------------------------------------------------------- reorient (ProprioceptiveEnvV0 / Geometries*)
def euler2quat(euler):
    """utils/quat_math.py:70-86 restated."""
    e = np.asarray(euler, np.float64)
    ai, aj, ak = e[2] / 2, -e[1] / 2, e[0] / 2
    si, sj, sk = np.sin(ai), np.sin(aj), np.sin(ak)
    ci, cj, ck = np.cos(ai), np.cos(aj), np.cos(ak)
    cc, cs, sc, ss = ci * ck, ci * sk, si * ck, si * sk
    return np.array([cj * cc + sj * ss, cj * cs - sj * sc, -(cj * ss + sj * cc), cj * sc - sj * cs])


def reorient_reset_draws(size_tables, env: int, episode: int, seed: int, tar_length: float, typed: bool = True):
    """(geom_type, size[3], axis_half, des_rot[3]) of the device-side reorient reset (k_reset, reorient branch), float32
    draws.  size_tables: [4][ntab][3] (capsule, ellipsoid, cylinder, box) when typed, else one [ntab][3] capsule table."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = philox4x32_10(0, 3, env, episode, k0, k1)
    u = [float(u01(x)) for x in c]
    tab = np.asarray(size_tables, np.float32)
    ntab = tab.shape[-2]
    idx = min(int(np.float32(u[0]) * np.float32(ntab)), ntab - 1)
    ty = min(int(np.float32(u[3]) * np.float32(4.0)), 3) if typed else 0
    row = tab[ty][idx] if typed else tab[idx]
    size = row.astype(np.float64)
    ah = float(np.float32(1.3) * row[1]) if ty == 0 else float(row[1] if ty == 2 else row[2])      # reorient_sar_v0.py:390-406
    e0 = -1.0 + 2.0 * u[1]; e1 = -0.8 + 2.0 * u[2]
    q = euler2quat([e0, e1, 0.0])
    w, x, y, z = q
    col = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    return 3 + ty, size, ah, col * 2 * ah / tar_length


def reorient_obs_reward(qpos, qvel, act, obj_xpos, obj_xmat, eps_pos, axis_half, des_rot, actuator_length,
                        actuator_velocity, actuator_force, dt, pen_length, rwd_keys_wt, obs_muscle=True):
    """get_obs_dict + obsdict2obsvec + get_reward_dict of reorient_sar_v0.py:116-174 (obs_muscle=False: pen_v0.py:88-169,
    the same arithmetic without the mlen / mvel / mforce observation blocks) on raw arrays."""
    R = np.asarray(obj_xmat, np.float64).reshape(3, 3)
    obj_rot = R[:, 2] * 2 * axis_half / pen_length          # (geom_xpos[top] - geom_xpos[bot]) / pen_length
    od = collections.OrderedDict(
        hand_jnt=qpos[:-6].copy(), obj_pos=np.asarray(obj_xpos, np.float64).copy(), obj_vel=qvel[-6:] * dt, obj_rot=obj_rot,
        obj_des_rot=np.asarray(des_rot, np.float64), obj_err_pos=obj_xpos - eps_pos, obj_err_rot=obj_rot - des_rot,
        mlen=actuator_length.copy(), mvel=actuator_velocity.copy(), mforce=actuator_force.copy(), act=act.copy())
    if not obs_muscle:
        for k in ("mlen", "mvel", "mforce"):
            del od[k]
    obs = np.concatenate([np.asarray(v, np.float64).ravel() for v in od.values()])
    pos_align = np.linalg.norm(od["obj_err_pos"])
    nrm = np.linalg.norm(obj_rot) * np.linalg.norm(des_rot)
    rot_align = float(np.dot(obj_rot, des_rot) / (nrm if nrm != 0 else 1.0))        # vector_math.py:10-34
    dropped = pos_align > 0.075
    na = act.size
    act_mag = np.linalg.norm(act) / na if na else 0.0
    rwd = collections.OrderedDict((
        ("pos_align", -1.0 * pos_align), ("rot_align", rot_align), ("act_reg", -1.0 * act_mag), ("drop", -1.0 * dropped),
        ("bonus", 1.0 * (rot_align > 0.9) * (pos_align < 0.075) + 5.0 * (rot_align > 0.95) * (pos_align < 0.075)),
        ("sparse", -1.0 * pos_align + rot_align), ("solved", (rot_align > 0.95) * (not dropped)), ("done", dropped)))
    rwd["dense"] = np.sum([wt * rwd[k] for k, wt in rwd_keys_wt.items()], axis=0)
    return obs, rwd


class ReorientEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of the reorient env on the fp64 oracle engine."""
    RWD_KEYS_WT = {"pos_align": 1.0, "rot_align": 1.0, "act_reg": 5.0, "drop": 5.0, "bonus": 10.0}

    def __init__(self, compiled, frame_skip=5, normalize_act=True, muscle_condition=""):
        super().__init__(compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        cm = compiled
        gp = cm.arrays["GEOM_POS"].reshape(-1, 3).astype(np.float64); g = cm.names["geom"]
        self.pen_length = float(np.linalg.norm(gp[g["top"]] - gp[g["bot"]]))
        self.tar_length = float(np.linalg.norm(gp[g["t_top"]] - gp[g["t_bot"]]))
        self.obj_b = cm.body_id("Object"); self.eps_s = cm.site_id("eps_ball"); self.obj_g = g["obj"]
        self.init_qpos = cm.qpos0.astype(np.float64).copy(); self.init_qpos[:-6] *= 0; self.init_qpos[0] = -1.5

    def reset(self, size, axis_half, des_rot, gtype: int = 3):
        self.d.reset()
        self.d.qpos[:] = self.init_qpos
        self.d.set_geom_size(self.obj_g, size, gtype)
        self.axis_half = float(axis_half); self.des_rot = np.asarray(des_rot, np.float64)
        self.steps = 0
        self.d.ctrl[:] = 0
        self.d.forward()
        return self._obs_rwd()[0]

    def _obs_rwd(self):
        d = self.d
        return reorient_obs_reward(d.qpos, d.qvel, d.act, d.xpos[self.obj_b], d.xmat[self.obj_b], d.site_xpos[self.eps_s],
                                   self.axis_half, self.des_rot, d.actuator_length, d.actuator_velocity, d.actuator_force,
                                   self.dt, self.pen_length, self.rwd_keys_wt)

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        self.d.ctrl[:] = ctrl
        self.d.step(self.frame_skip)
        self.d.forward()
        obs, rwd = self._obs_rwd()
        self.steps += 1
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd


def pen_reset_draws(env: int, episode: int, seed: int, axis_half: float, tar_length: float, ranges=(-1.0, 1.0, -1.0, 1.0)):
    """des_rot[3] of the device-side pen reset (k_reset, pen branch): words 1 / 2 of counter (0, 3, env, episode)."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = philox4x32_10(0, 3, env, episode, k0, k1)
    u = [float(u01(x)) for x in c]
    e0 = ranges[0] + (ranges[1] - ranges[0]) * u[1]; e1 = ranges[2] + (ranges[3] - ranges[2]) * u[2]
    w, x, y, z = euler2quat([e0, e1, 0.0])
    col = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    return col * 2 * axis_half / tar_length


class PenTwirlEnvOracle(ReorientEnvOracle):
    """Single-env CPU restatement of PenTwirl{Fixed,Random}EnvV0 (pen_v0.py) on the fp64 oracle engine."""

    def __init__(self, compiled, frame_skip=5, normalize_act=True, muscle_condition=""):
        PoseEnvOracle.__init__(self, compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        cm = compiled
        sp = cm.arrays["SITE_POS"].reshape(-1, 3).astype(np.float64); sn = cm.names["site"]
        self.pen_length = float(np.linalg.norm(sp[sn["object_top"]] - sp[sn["object_bottom"]]))
        self.tar_length = float(np.linalg.norm(sp[sn["target_top"]] - sp[sn["target_bottom"]]))
        self.obj_b = cm.body_id("Object"); self.eps_s = cm.site_id("eps_ball")
        self.init_qpos = cm.qpos0.astype(np.float64).copy(); self.init_qpos[:-6] *= 0; self.init_qpos[0] = -1.5
        self.axis_half = 0.5 * self.pen_length

    def reset(self, des_rot):
        self.d.reset()
        self.d.qpos[:] = self.init_qpos
        self.des_rot = np.asarray(des_rot, np.float64)
        self.steps = 0
        self.d.ctrl[:] = 0
        self.d.forward()
        return self._obs_rwd()[0]

    def _obs_rwd(self):
        d = self.d
        return reorient_obs_reward(d.qpos, d.qvel, d.act, d.xpos[self.obj_b], d.xmat[self.obj_b], d.site_xpos[self.eps_s],
                                   self.axis_half, self.des_rot, d.actuator_length, d.actuator_velocity, d.actuator_force,
                                   self.dt, self.pen_length, self.rwd_keys_wt, obs_muscle=False)


# ---------------------------------------------------------------------- ObjHold (obj_hold_v0.py)
def objhold_reset_draws(env: int, episode: int, seed: int, center, goal_half: float, size_range=None):
    """(goal[3], size[3] | None) of the device-side object-hold reset (k_reset, hold branch), float32 draws."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    c = philox4x32_10(0, 4, env, episode, k0, k1)
    u = np.array([u01(x) for x in c[:3]], np.float32).reshape(3)
    goal = (np.asarray(center, np.float32) + np.float32(goal_half) * (np.float32(2.0) * u - np.float32(1.0))).astype(np.float32)
    size = None
    if size_range is not None:
        c2 = philox4x32_10(1, 4, env, episode, k0, k1)
        u2 = np.array([u01(x) for x in c2[:3]], np.float32).reshape(3)
        size = (np.float32(size_range[0]) + np.float32(size_range[1] - size_range[0]) * u2).astype(np.float32)
    return goal, size


def objhold_obs_reward(qpos, qvel, act, obj_pos, goal_pos, dt, rwd_keys_wt):
    """get_obs_dict + obsdict2obsvec + get_reward_dict of obj_hold_v0.py:82-131 on raw arrays."""
    od = collections.OrderedDict(hand_qpos=qpos[:-7].copy(), hand_qvel=qvel[:-6] * dt, obj_pos=np.asarray(obj_pos, np.float64),
                                 obj_err=np.asarray(goal_pos, np.float64) - obj_pos, act=act.copy())
    obs = np.concatenate([np.asarray(v, np.float64).ravel() for v in od.values()])
    goal_dist = np.abs(np.linalg.norm(od["obj_err"]))
    na = act.size
    act_mag = np.linalg.norm(act) / na if na else 0.0
    gaol_th = 0.010
    drop = goal_dist > 0.300
    rwd = collections.OrderedDict((
        ("goal_dist", -1.0 * goal_dist), ("bonus", 1.0 * (goal_dist < 2 * gaol_th) + 1.0 * (goal_dist < gaol_th)),
        ("act_reg", -1.0 * act_mag), ("penalty", -1.0 * drop), ("sparse", -goal_dist), ("solved", goal_dist < gaol_th), ("done", drop)))
    rwd["dense"] = np.sum([wt * rwd[k] for k, wt in rwd_keys_wt.items()], axis=0)
    return obs, rwd


class ObjHoldEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of ObjHold{Fixed,Random}EnvV0 on the fp64 oracle engine."""
    RWD_KEYS_WT = {"goal_dist": 100.0, "bonus": 4.0, "penalty": 10}

    def __init__(self, compiled, frame_skip=10, normalize_act=True, muscle_condition=""):
        super().__init__(compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        cm = compiled
        self.obj_s = cm.site_id("object"); self.obj_g = cm.names["geom"]["object"]
        self.init_qpos = cm.qpos0.astype(np.float64).copy(); self.init_qpos[:-7] *= 0; self.init_qpos[0] = -1.5

    def reset(self, goal, size=None):
        self.d.reset()
        self.d.qpos[:] = self.init_qpos
        if size is not None:
            self.d.set_geom_size(self.obj_g, size)
        self.goal = np.asarray(goal, np.float64)
        self.steps = 0
        self.d.ctrl[:] = 0
        self.d.forward()
        return self._obs_rwd()[0]

    def _obs_rwd(self):
        d = self.d
        return objhold_obs_reward(d.qpos, d.qvel, d.act, d.site_xpos[self.obj_s], self.goal, self.dt, self.rwd_keys_wt)

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        self.d.ctrl[:] = ctrl
        self.d.step(self.frame_skip)
        self.d.forward()
        obs, rwd = self._obs_rwd()
        self.steps += 1
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd


# ---------------------------------------------------------------------- KeyTurn (key_turn_v0.py)
def keyturn_obs_reward(qpos, qvel, act, keyhead, iftip, thtip, dt, goal_th, rwd_keys_wt):
    """get_obs_dict + obsdict2obsvec + get_reward_dict of key_turn_v0.py:101-150 on raw arrays."""
    od = collections.OrderedDict(hand_qpos=qpos[:-1].copy(), hand_qvel=qvel[:-1] * dt, key_qpos=np.array([qpos[-1]]),
                                 key_qvel=np.array([qvel[-1]]) * dt, IFtip_approach=np.asarray(keyhead, np.float64) - iftip,
                                 THtip_approach=np.asarray(keyhead, np.float64) - thtip, act=act.copy())
    obs = np.concatenate([np.asarray(v, np.float64).ravel() for v in od.values()])
    IF_d = np.abs(np.linalg.norm(od["IFtip_approach"]) - 0.030)
    TH_d = np.abs(np.linalg.norm(od["THtip_approach"]) - 0.030)
    key_pos = od["key_qpos"][0]
    na = act.size
    act_mag = np.linalg.norm(act) / na if na else 0.0
    far_th = 0.1
    rwd = collections.OrderedDict((
        ("key_turn", key_pos), ("IFtip_approach", -1.0 * IF_d), ("THtip_approach", -1.0 * TH_d), ("act_reg", -1.0 * act_mag),
        ("bonus", 1.0 * (key_pos > np.pi / 2) + 1.0 * (key_pos > np.pi)),
        ("penalty", -1.0 * (IF_d > far_th / 2) - 1.0 * (TH_d > far_th / 2)),
        ("sparse", key_pos), ("solved", key_pos > goal_th), ("done", (IF_d > far_th) or (TH_d > far_th))))
    rwd["dense"] = np.sum([wt * rwd[k] for k, wt in rwd_keys_wt.items()], axis=0)
    return obs, rwd


class KeyTurnEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of KeyTurnEnvV0 on the fp64 oracle engine."""
    RWD_KEYS_WT = {"key_turn": 1.0, "IFtip_approach": 10.0, "THtip_approach": 10.0, "act_reg": 1.0, "bonus": 4.0, "penalty": 25.0}

    def __init__(self, compiled, goal_th=3.14, frame_skip=10, normalize_act=True, muscle_condition=""):
        super().__init__(compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        cm = compiled
        self.goal_th = goal_th
        self.kh, self.IF, self.TH = cm.site_id("keyhead"), cm.site_id("IFtip"), cm.site_id("THtip")
        self.init_qpos = cm.qpos0.astype(np.float64).copy(); self.init_qpos[:-1] *= 0

    def reset(self, key_q0, key_pos=None):
        self.d.reset()
        self.d.qpos[:] = self.init_qpos
        self.d.qpos[-1] = key_q0
        if key_pos is not None:
            self.d.set_body_pos(self.cm.nbody - 1, key_pos)
        self.steps = 0
        self.d.ctrl[:] = 0
        self.d.forward()
        return self._obs_rwd()[0]

    def _obs_rwd(self):
        d = self.d
        return keyturn_obs_reward(d.qpos, d.qvel, d.act, d.site_xpos[self.kh], d.site_xpos[self.IF], d.site_xpos[self.TH],
                                  self.dt, self.goal_th, self.rwd_keys_wt)

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        self.d.ctrl[:] = ctrl
        self.d.step(self.frame_skip)
        self.d.forward()
        obs, rwd = self._obs_rwd()
        self.steps += 1
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd


# ---------------------------------------------------------------------- leg stand (walk_v0.py ReachEnvV0)
def stand_obs_reward(qpos, qvel, act, tip_pos, target_pos, time, dt, far_th, rwd_keys_wt):
    """get_obs_dict + obsdict2obsvec + get_reward_dict of walk_v0.py:71-128 on raw arrays (ntip tips concatenated)."""
    tip_pos = np.asarray(tip_pos, np.float64).ravel(); target_pos = np.asarray(target_pos, np.float64).ravel()
    ntip = tip_pos.size // 3
    od = collections.OrderedDict(qpos=qpos.copy(), qvel=qvel * dt, tip_pos=tip_pos, reach_err=target_pos - tip_pos, act=act.copy())
    obs = np.concatenate([np.asarray(v, np.float64).ravel() for v in od.values()])
    reach_dist = np.linalg.norm(od["reach_err"])
    vel_dist = np.linalg.norm(od["qvel"])
    na = act.size
    act_mag = np.linalg.norm(act) / na if na else 0.0
    fth = far_th * ntip if time > 2 * dt else np.inf
    near_th = ntip * 0.050
    rwd = collections.OrderedDict((
        ("reach", 10.0 - 1.0 * reach_dist - 10.0 * vel_dist),
        ("bonus", 1.0 * (reach_dist < 2 * near_th) + 1.0 * (reach_dist < near_th)),
        ("act_reg", -100.0 * act_mag), ("penalty", -1.0 * (reach_dist > fth)), ("sparse", -1.0 * reach_dist),
        ("solved", reach_dist < near_th), ("done", reach_dist > fth)))
    rwd["dense"] = np.sum([wt * rwd[k] for k, wt in rwd_keys_wt.items()], axis=0)
    return obs, rwd


def stand_generate_qpos(init_qpos, jnt_qposadr, jnt_range, draw):
    """walk_v0.py:153-168: init + draw on the jnt_qposadr entries, clipped to jnt_range."""
    q = np.asarray(init_qpos, np.float64).copy()
    q[jnt_qposadr] += np.asarray(draw)[jnt_qposadr]
    q[jnt_qposadr] = np.clip(q[jnt_qposadr], jnt_range[:, 0], jnt_range[:, 1])
    return q


class StandEnvOracle(PoseEnvOracle):
    """Single-env CPU restatement of walk_v0.ReachEnvV0 (myoLegStandRandom-v0) on the fp64 oracle engine."""
    RWD_KEYS_WT = {"reach": 1.0, "bonus": 4.0, "penalty": 50, "act_reg": 1}

    def __init__(self, compiled, tip_sids, far_th, frame_skip=10, normalize_act=True, muscle_condition=""):
        super().__init__(compiled, 0.0, frame_skip, normalize_act, muscle_condition, dict(self.RWD_KEYS_WT))
        self.tip_sids = list(tip_sids); self.far_th = far_th
        self.target_pos = np.zeros(3 * len(self.tip_sids))

    def tip_pos(self):
        return np.concatenate([self.d.site_xpos[s] for s in self.tip_sids])

    def place(self, qpos, qvel=None):
        self.d.reset()
        self.d.qpos[:] = qpos
        if qvel is not None:
            self.d.qvel[:] = qvel
        self.d.ctrl[:] = 0
        self.d.forward()
        self.steps = 0

    def _obs_rwd(self):
        d = self.d
        return stand_obs_reward(d.qpos, d.qvel, d.act, self.tip_pos(), self.target_pos, d.time, self.dt, self.far_th, self.rwd_keys_wt)

    def step(self, a):
        a = np.asarray(a, np.float64)
        ctrl = a.copy()
        if self.cm.na and self.normalize_act:
            ctrl[self.muscle] = 1.0 / (1.0 + np.exp(-5.0 * (ctrl[self.muscle] - 0.5)))
        self.d.ctrl[:] = ctrl
        self.d.step(self.frame_skip)
        self.d.forward()
        obs, rwd = self._obs_rwd()
        self.steps += 1
        self.rwd_dict = rwd
        return obs, float(rwd["dense"]), bool(rwd["done"]), rwd
