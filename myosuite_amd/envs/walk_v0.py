"""Batched WalkEnvV0 -- host-side mirror of myosuite/envs/myo/myobase/walk_v0.py:189-540.

obs keys ``qpos_without_xy, qvel, com_vel, torso_angle, feet_heights, height, feet_rel_positions, phase_var,
muscle_length, muscle_velocity, muscle_force`` (+ ``act``), reward keys ``vel_reward, done, cyclic_hip, ref_rot,
joint_angle_rew``; ``self.steps`` of the reference is the device-side ``step_count``.  The whole env-step (ctrl map /
fatigue, frame_skip x mj_step with foot-ground contacts and knee equalities, final forward, obs, reward) is one fused
kernel launch; reset draws (keyframe coin + N(0, 0.02) noise, walk_v0.py:327-352) are Philox-keyed on the device.

Not covered: the terrain variants (``TerrainEnvV0``, walk_v0.py:543-680) need height-field collision.
"""
from __future__ import annotations

import collections
from typing import Optional

import numpy as np
import torch

from .. import engine as E
from .base_v0 import BaseV0
from .spaces import Box


class WalkEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["qpos_without_xy", "qvel", "com_vel", "torso_angle", "feet_heights", "height",
                        "feet_rel_positions", "phase_var", "muscle_length", "muscle_velocity", "muscle_force"]   # walk_v0.py:191-203
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"vel_reward": 5.0, "done": -100, "cyclic_hip": -10, "ref_rot": 10.0,
                                    "joint_angle_rew": 5.0}                                                        # walk_v0.py:205-211

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=1000,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, obs_keys=DEFAULT_OBS_KEYS, weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS, min_height=0.8,
               max_rot=0.8, hip_period=100, reset_type="init", target_x_vel=0.0, target_y_vel=1.2, target_rot=None,
               **kwargs):
        self.min_height, self.max_rot, self.hip_period = float(min_height), float(max_rot), int(hip_period)
        self.reset_type = reset_type
        self.target_x_vel, self.target_y_vel, self.target_rot = float(target_x_vel), float(target_y_vel), target_rot
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        self.init_qpos = cm.key_qpos[0].astype(np.float32).copy()      # walk_v0.py:271-272
        self.init_qvel = np.zeros(cm.nv, np.float32)
        self._keys_q = [torch.from_numpy(cm.key_qpos[k].astype(np.float32)).to(dev) for k in range(cm.key_qpos.shape[0])]
        self._keys_v = [torch.from_numpy(cm.key_qvel[k].astype(np.float32)).to(dev) for k in range(cm.key_qvel.shape[0])]
        nu = cm.nu
        self.obs_dim = (cm.nq - 2) + cm.nv + 2 + 4 + 2 + 1 + 6 + 1 + 3 * nu + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_WALK), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_WALK)
        for i, b in enumerate(("pelvis", "torso", "talus_l", "talus_r")):
            t.walk_body[i] = cm.body_id(b)
        qadr = cm.arrays["JNT_QPOSADR"]
        for i, j in enumerate(("hip_flexion_l", "hip_flexion_r", "hip_adduction_l", "hip_adduction_r", "hip_rotation_l",
                               "hip_rotation_r")):
            t.walk_qadr[i] = int(qadr[cm.joint_id(j)])
        t.walk_min_height, t.walk_max_rot, t.walk_hip_period = self.min_height, self.max_rot, self.hip_period
        t.walk_target_x_vel, t.walk_target_y_vel = self.target_x_vel, self.target_y_vel
        rot = self.target_rot if self.target_rot is not None else self.init_qpos[3:7]
        for i in range(4):
            t.walk_target_rot[i] = float(rot[i])
        self._check_reward_keys(("vel_reward", "done", "cyclic_hip", "ref_rot", "joint_angle_rew"))
        for i, k in enumerate(("vel_reward", "done", "cyclic_hip", "ref_rot", "joint_angle_rew")):
            t.walk_w[i] = float(w.get(k, 0.0))
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    @property
    def steps(self) -> torch.Tensor:            # walk_v0.py:256,355-358
        return self.step_count

    def _refresh_dicts(self):
        cm = self.cm
        o = self.obs
        sizes = [("qpos_without_xy", cm.nq - 2), ("qvel", cm.nv), ("com_vel", 2), ("torso_angle", 4), ("feet_heights", 2),
                 ("height", 1), ("feet_rel_positions", 6), ("phase_var", 1), ("muscle_length", cm.nu),
                 ("muscle_velocity", cm.nu), ("muscle_force", cm.nu), ("act", cm.na)]
        od = collections.OrderedDict(t=self.state.time, time=self.state.time)
        k0 = 0
        for k, sz in sizes:
            od[k] = o[:, k0:k0 + sz]; k0 += sz
        self.obs_dict = od
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_WALK))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def _rollout_fill_reset(self, ro):
        """the walk reset (key pose, optional stride coin + noise; 3CC-r state back to rest) runs inside the env-step launch
        where an env is a whole wavefront (mm_rollout: the launch makes a second, reset-observation pass for the envs it
        re-arms); random fatigue resets and narrower launches keep the separate reset calls"""
        if not (self.autoreset and self.hm.info(E.INFO_FOLDED_RESET) == 1 and not self.fatigue_reset_random):
            return
        rnd = self.reset_type == "random"
        k = 2 if (rnd or self.reset_type == "init") else 0
        ro.autoreset = 1
        ro.walk_ka_qpos, ro.walk_ka_qvel = self._keys_q[k].data_ptr(), self._keys_v[k].data_ptr()
        ro.walk_kb_qpos = self._keys_q[3].data_ptr() if rnd else None
        ro.walk_kb_qvel = self._keys_v[3].data_ptr() if rnd else None
        ro.walk_random = int(rnd)
        ro.episode = self.episode.data_ptr(); ro.reset_seed = self._seed_u64
        if self.muscle_condition == "fatigue" and self.fatigue_reset_vec is not None:
            self._ro_fat_vec = torch.from_numpy(np.broadcast_to(np.asarray(self.fatigue_reset_vec, np.float32), (self.cm.na,)).copy()).to(self.device)
            ro.fat_reset_vec = self._ro_fat_vec.data_ptr()

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        if self.reset_type == "random":          # walk_v0.py:327-352
            E.walk_reset(self.hm, self.state, mask, self._keys_q[2], self._keys_v[2], self._keys_q[3], self._keys_v[3],
                         True, self.episode, self.step_count, self._seed_u64)
        else:
            k = 2 if self.reset_type == "init" else 0
            E.walk_reset(self.hm, self.state, mask, self._keys_q[k], self._keys_v[k], None, None, False, self.episode,
                         self.step_count, self._seed_u64)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}
