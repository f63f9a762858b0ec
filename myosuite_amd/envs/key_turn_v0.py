"""Batched KeyTurnEnvV0 -- host-side mirror of myosuite/envs/myo/myobase/key_turn_v0.py:14-170.

obs keys ``hand_qpos, hand_qvel, key_qpos, key_qvel, IFtip_approach, THtip_approach`` (+ ``act``) = 93; reward keys
``key_turn, IFtip_approach, THtip_approach, act_reg, bonus, penalty``.  The key's hinge carries dry friction (MuJoCo
``frictionloss``: friction-loss constraint rows in the fused kernel).  Reset: hand fully open, key angle ~ U(key_init_range);
the Random task also moves the key body by U(-1 cm, 1 cm)^3 per episode (per-env model delta ``mm_state.body_pos_env``).
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


class KeyTurnEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["hand_qpos", "hand_qvel", "key_qpos", "key_qvel", "IFtip_approach", "THtip_approach"]   # key_turn_v0.py:16-23
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"key_turn": 1.0, "IFtip_approach": 10.0, "THtip_approach": 10.0, "act_reg": 1.0,
                                    "bonus": 4.0, "penalty": 25.0}                                              # key_turn_v0.py:24-31

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=200,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, goal_th: float = 3.14, obs_keys=DEFAULT_OBS_KEYS, weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS,
               key_init_range: tuple = (0, 0), **kwargs):
        self.goal_th = float(goal_th)
        self.key_init_range = (float(key_init_range[0]), float(key_init_range[1]))
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        self.keyhead_sid, self.IF_sid, self.TH_sid = cm.site_id("keyhead"), cm.site_id("IFtip"), cm.site_id("THtip")
        km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
        sx = km.site_xpos(km.fk(cm.qpos0.astype(np.float64)[None]))[0]
        self.key_init_pos = sx[self.keyhead_sid].astype(np.float32)             # key_turn_v0.py:65 (setup-time forward)
        self.init_qpos = cm.qpos0.astype(np.float32).copy()
        self.init_qpos[:-1] *= 0                                                # key_turn_v0.py:72: fully open hand
        self.randomize = self.key_init_range[0] != self.key_init_range[1]       # key_turn_v0.py:163
        self._sites = torch.tensor([self.keyhead_sid, self.IF_sid, self.TH_sid], dtype=torch.int32, device=dev)
        self._klo = torch.tensor([self.key_init_range[0]], **f); self._khi = torch.tensor([self.key_init_range[1]], **f)
        self.key_q0 = torch.zeros(n, 1, **f)
        if self.randomize:
            self.key_body = cm.nbody - 1                                        # body_pos[-1]
            self.body_pos = torch.from_numpy(np.tile(self.key_init_pos, (n, 1))).to(dev).contiguous()
            self.state.set_body_pos_env(self.key_body, self.body_pos)
            self._plo = torch.full((3,), -0.01, **f); self._phi = torch.full((3,), 0.01, **f)
            self._pbase = torch.from_numpy(self.key_init_pos).to(dev)
        self._init_q = torch.from_numpy(self.init_qpos).to(dev)
        self.obs_dim = (cm.nq - 1) + (cm.nv - 1) + 2 + 6 + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_KEYTURN), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_KEYTURN)
        t.tip_sites = self._sites.data_ptr(); t.ntip = 3
        t.key_goal_th = self.goal_th
        self._check_reward_keys(("key_turn", "IFtip_approach", "THtip_approach", "act_reg", "bonus", "penalty"))
        for i, k in enumerate(("key_turn", "IFtip_approach", "THtip_approach", "act_reg", "bonus", "penalty")):
            t.key_w[i] = float(w.get(k, 0.0))
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _refresh_dicts(self):
        cm = self.cm
        nh, nhv = cm.nq - 1, cm.nv - 1
        o = self.obs
        k = nh + nhv
        self.obs_dict = collections.OrderedDict(
            time=self.state.time, hand_qpos=o[:, :nh], hand_qvel=o[:, nh:k], key_qpos=o[:, k:k + 1], key_qvel=o[:, k + 1:k + 2],
            IFtip_approach=o[:, k + 2:k + 5], THtip_approach=o[:, k + 5:k + 8], act=o[:, k + 8:])
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((kk, r[:, i]) for i, kk in enumerate(E.RWD_KEYS_KEYTURN))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        """key_turn_v0.py:155-170.  Draws are Philox streams of (seed, env, episode): 17 = key angle, 18 = key position."""
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        E.env_draw(self.key_q0, self._klo, self._khi, mask, self.episode, self._seed_u64, 17, env_index_base=self.env_index_base)
        if self.randomize:
            E.env_draw(self.body_pos, self._plo, self._phi, mask, self.episode, self._seed_u64, 18, base=self._pbase, env_index_base=self.env_index_base)
        q = self._init_q.expand(self.num_envs, -1).clone()
        q[:, -1] = self.key_q0[:, 0]
        E.reset(self.hm, self.state, mask, q.contiguous(), None)
        if mask is None:
            self.episode += 1; self.step_count.zero_()
        else:
            m = mask.bool()
            self.episode += m.to(torch.int32)
            self.step_count.masked_fill_(m, 0)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}
