"""Batched ReachEnvV0 -- host-side mirror of myosuite/envs/myo/myobase/reach_v0.py:15-172.

obs keys ``qpos, qvel, tip_pos, reach_err`` (+ ``act``), reward keys ``reach, bonus, penalty`` (+ act_reg), targets are
the world positions of the ``*_target`` sites, re-drawn at every reset.  The fused kernel runs the post-step forward pass
(the task reads ``site_xpos``) and writes obs/reward.

Synthetic-model note: the reference's target boxes are absolute coordinates of the real myoHand scene
(myobase/__init__.py:521-575).  Our synthetic hand lives elsewhere, so each box is re-centred on the synthetic tip position
at qpos0 while keeping the reference's spans (``target_center`` kwarg carries the reference centres).  The synthetic
finger is laid out in the reference's coordinates, so its boxes are used as they are (``target_frame="world"``).
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


class ReachEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["qpos", "qvel", "tip_pos", "reach_err"]                    # reach_v0.py:17
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"reach": 1.0, "bonus": 4.0, "penalty": 50}     # reach_v0.py:18-22

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=100,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, target_reach_range: dict, far_th=0.35, obs_keys=DEFAULT_OBS_KEYS,
               weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS, target_center: Optional[dict] = None,
               target_frame: str = "tip", **kwargs):
        self.far_th = float(far_th)
        self.target_reach_range = target_reach_range
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys,
                       sites=list(target_reach_range.keys()), **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        self.ntip = len(self.tip_sids)
        # tip positions at qpos0 (host model-compiler kinematics; setup time only)
        km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
        sx = km.site_xpos(km.fk(cm.qpos0.astype(np.float64)[None]))[0]
        tip0 = sx[self.tip_sids]
        lo, hi = [], []
        for i, (site, span) in enumerate(target_reach_range.items()):
            span = np.asarray(span, np.float64)
            if target_frame == "world":       # the reference's absolute coordinates (models laid out like the reference's)
                lo.append(span[0]); hi.append(span[1])
                continue
            c = np.asarray(target_center[site], np.float64) if target_center else 0.5 * (span[0] + span[1])
            lo.append(tip0[i] + (span[0] - c)); hi.append(tip0[i] + (span[1] - c))
        self._tlo = torch.from_numpy(np.concatenate(lo).astype(np.float32)).to(dev)
        self._thi = torch.from_numpy(np.concatenate(hi).astype(np.float32)).to(dev)
        self._tip0 = torch.from_numpy(tip0.reshape(-1).astype(np.float32)).to(dev)
        self._tip_sites = torch.tensor(self.tip_sids, dtype=torch.int32, device=dev)
        self.target_pos = torch.zeros(n, 3 * self.ntip, **f)
        self.obs_dim = cm.nq + cm.nv + 6 * self.ntip + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_REACH), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_REACH)
        self._check_reward_keys(("reach", "bonus", "act_reg", "penalty"))
        t.w_pose = float(w.get("reach", 0.0)); t.w_bonus = float(w.get("bonus", 0.0))
        t.w_act_reg = float(w.get("act_reg", 0.0)); t.w_penalty = float(w.get("penalty", 0.0))
        t.tip_sites = self._tip_sites.data_ptr(); t.ntip = self.ntip; t.target_pos = self.target_pos.data_ptr()
        t.reach_far_th = self.far_th
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _refresh_dicts(self):
        cm = self.cm
        nq, nv, na, n3 = cm.nq, cm.nv, cm.na, 3 * self.ntip
        o = self.obs
        self.obs_dict = collections.OrderedDict(
            time=self.state.time, qpos=o[:, :nq], qvel=o[:, nq:nq + nv], tip_pos=o[:, nq + nv:nq + nv + n3],
            target_pos=self.target_pos, reach_err=o[:, nq + nv + n3:nq + nv + 2 * n3], act=o[:, nq + nv + 2 * n3:])
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_REACH))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        E.reach_reset(self.hm, self.state, mask, self._tlo, self._thi, self.target_pos, self._tip0, self.ntip,
                      self.episode, self.step_count, self._seed_u64, obs=self.obs)
        self._refresh_dicts()
        return self._obs_out(), {}
