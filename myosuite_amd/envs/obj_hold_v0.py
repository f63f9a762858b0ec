"""Batched ObjHold{Fixed,Random}EnvV0 -- host-side mirror of myosuite/envs/myo/myobase/obj_hold_v0.py:13-145.

obs keys ``hand_qpos, hand_qvel, obj_pos, obj_err`` (+ ``act``) = 91, reward keys ``goal_dist, bonus, penalty`` (+ act_reg);
the object is a free-floating ellipsoid (frictionless contacts against the hand capsules).  The Random task re-draws the goal
position around the object's initial position and the object's size every episode (per-env model delta
``mm_state.geom_size_env``); the reference also copies the size to the goal site for rendering, which has no physics effect.
"""
from __future__ import annotations

import collections
from typing import Optional

import numpy as np
import torch

from .. import engine as E
from ..model import kin_np as K
from .base_v0 import BaseV0
from .spaces import Box


class ObjHoldEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["hand_qpos", "hand_qvel", "obj_pos", "obj_err"]                        # obj_hold_v0.py:15
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"goal_dist": 100.0, "bonus": 4.0, "penalty": 10}             # obj_hold_v0.py:16-20

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=75,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, randomize: bool = False, obs_keys=DEFAULT_OBS_KEYS, weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS,
               **kwargs):
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        self.randomize = bool(randomize)
        self.object_sid, self.goal_sid = cm.site_id("object"), cm.site_id("goal")
        km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
        sx = km.site_xpos(km.fk(cm.qpos0.astype(np.float64)[None]))[0]
        self.object_init_pos = sx[self.object_sid].astype(np.float32)                           # obj_hold_v0.py:57 (setup-time forward)
        goal_model = sx[self.goal_sid].astype(np.float32)
        self.init_qpos = cm.qpos0.astype(np.float32).copy()
        self.init_qpos[:-7] *= 0; self.init_qpos[0] = -1.5                                      # obj_hold_v0.py:64-65
        self._init_qpos_dev = torch.from_numpy(self.init_qpos).to(dev)
        self._goal_center = torch.from_numpy(self.object_init_pos if self.randomize else goal_model).to(dev)
        self.goal = torch.zeros(n, 3, **f)
        self._obj_site = torch.tensor([self.object_sid], dtype=torch.int32, device=dev)
        if self.randomize:
            self.geom_size = torch.zeros(n, 3, **f)
            self.state.set_geom_size_env(cm.names["geom"]["object"], self.geom_size)
        self.obs_dim = (cm.nq - 7) + (cm.nv - 6) + 3 + 3 + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_OBJHOLD), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_OBJHOLD)
        self._check_reward_keys(("goal_dist", "bonus", "act_reg", "penalty"))
        t.w_pose = float(w.get("goal_dist", 0.0)); t.w_bonus = float(w.get("bonus", 0.0))
        t.w_act_reg = float(w.get("act_reg", 0.0)); t.w_penalty = float(w.get("penalty", 0.0))
        t.tip_sites = self._obj_site.data_ptr(); t.ntip = 1; t.target_pos = self.goal.data_ptr()
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _refresh_dicts(self):
        cm = self.cm
        nh, nhv = cm.nq - 7, cm.nv - 6
        o = self.obs
        self.obs_dict = collections.OrderedDict(
            time=self.state.time, hand_qpos=o[:, :nh], hand_qvel=o[:, nh:nh + nhv], obj_pos=o[:, nh + nhv:nh + nhv + 3],
            obj_err=o[:, nh + nhv + 3:nh + nhv + 6], act=o[:, nh + nhv + 6:])
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_OBJHOLD))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        E.objhold_reset(self.hm, self.state, mask, self._init_qpos_dev, self._goal_center, 0.030 if self.randomize else 0.0,
                        (0.020, 0.030) if self.randomize else None, self.goal, self.episode, self.step_count, self._seed_u64)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}
