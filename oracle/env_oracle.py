"""numpy/fp64 restatement of the reference's env-level arithmetic around the physics step.

TEST INFRASTRUCTURE ONLY (see oracle/mmo_engine.c header): used by tests/, smoke() and
bench.py's cpu_baseline leg.  Each function cites the reference lines it follows.
"""
from __future__ import annotations

import collections
import math

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
        return np.concatenate([od[k].ravel() for k in ("qpos", "qvel", "pose_err", "act")]).astype(np.float32)

    def get_reward_dict(self, od):                               # pose_v0.py:113-140
        pose_dist = np.linalg.norm(od["pose_err"], axis=-1)
        act_mag = np.linalg.norm(od["act"], axis=-1)
        if self.cm.na != 0:
            act_mag = act_mag / self.cm.na
        far_th = 4 * np.pi / 2
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
        return np.concatenate([od[k].ravel() for k in ("qpos", "qvel", "tip_pos", "reach_err", "act")]).astype(np.float32)

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
