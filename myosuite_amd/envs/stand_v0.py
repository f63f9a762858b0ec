"""Batched leg-stand env -- host-side mirror of ``ReachEnvV0`` in myosuite/envs/myo/myobase/walk_v0.py:15-186
(``myoLegStandRandom-v0``): keep the pelvis site on its target while standing.

obs keys ``qpos, qvel, tip_pos, reach_err`` (+ ``act``); reward keys ``reach`` (= 10 - reach_dist - 10 ||qvel dt||), ``bonus``,
``act_reg`` (x100), ``penalty``.  Reset (walk_v0.py:147-186): joints = keyframe 0 + U(joint_random_range), clipped to
``jnt_range`` -- including the (0, 0) range of unlimited joints, as the reference does; the target is the tip position at a
FIRST random pose + U(span), the episode then starts from a SECOND random pose.
"""
from __future__ import annotations

import collections
from typing import Optional

import numpy as np
import torch

from .. import engine as E
from .base_v0 import BaseV0
from .spaces import Box


class StandEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["qpos", "qvel", "tip_pos", "reach_err"]                                   # walk_v0.py:18
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"reach": 1.0, "bonus": 4.0, "penalty": 50, "act_reg": 1}      # walk_v0.py:19-24

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=150,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, target_reach_range: dict, joint_random_range: tuple = (0.0, 0.0), far_th=0.35, obs_keys=DEFAULT_OBS_KEYS,
               weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS, **kwargs):
        self.far_th = float(far_th)
        self.target_reach_range = target_reach_range
        self.joint_random_range = (float(joint_random_range[0]), float(joint_random_range[1]))
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys,
                       sites=list(target_reach_range.keys()), **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        self.ntip = len(self.tip_sids)
        self.init_qpos = cm.key_qpos[0].astype(np.float32).copy()                                 # walk_v0.py:62-63
        self.init_qvel = cm.key_qvel[0].astype(np.float32).copy()
        adr = cm.arrays["JNT_QPOSADR"].astype(np.int64)
        jr = cm.jnt_range.astype(np.float32)
        lo = np.zeros(cm.nq, np.float32); hi = np.zeros(cm.nq, np.float32); sel = np.zeros(cm.nq, np.float32)
        lo[adr] = jr[:, 0]; hi[adr] = jr[:, 1]; sel[adr] = 1.0                                    # walk_v0.py:156-166: jnt_qposadr entries only
        self._sel = torch.from_numpy(sel).to(dev); self._clo = torch.from_numpy(lo).to(dev); self._chi = torch.from_numpy(hi).to(dev)
        self._q0 = torch.from_numpy(self.init_qpos).to(dev); self._v0 = torch.from_numpy(self.init_qvel).to(dev)
        self._jlo = torch.full((cm.nq,), self.joint_random_range[0], **f); self._jhi = torch.full((cm.nq,), self.joint_random_range[1], **f)
        span = [np.asarray(s_, np.float32) for s_ in target_reach_range.values()]
        self._slo = torch.from_numpy(np.concatenate([s_[0] for s_ in span])).to(dev)
        self._shi = torch.from_numpy(np.concatenate([s_[1] for s_ in span])).to(dev)
        self._tip_sites = torch.tensor(self.tip_sids, dtype=torch.int32, device=dev)
        self.target_pos = torch.zeros(n, 3 * self.ntip, **f)
        self._draw_q = torch.zeros(n, cm.nq, **f); self._draw_t = torch.zeros(n, 3 * self.ntip, **f)
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
        t.reach_far_th = self.far_th; t.reach_stand = 1
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _refresh_dicts(self):
        cm = self.cm
        nq, nv, n3 = cm.nq, cm.nv, 3 * self.ntip
        o = self.obs
        self.obs_dict = collections.OrderedDict(
            time=self.state.time, qpos=o[:, :nq], qvel=o[:, nq:nq + nv], tip_pos=o[:, nq + nv:nq + nv + n3],
            target_pos=self.target_pos, reach_err=o[:, nq + nv + n3:nq + nv + 2 * n3], act=o[:, nq + nv + 2 * n3:])
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_REACH))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def _generate_qpos(self, mask, stream_id: int) -> torch.Tensor:
        """walk_v0.py:153-168 (Philox stream `stream_id` of (seed, env, episode))."""
        E.env_draw(self._draw_q, self._jlo, self._jhi, mask, self.episode, self._seed_u64, stream_id, env_index_base=self.env_index_base)
        q = self._q0 + self._sel * self._draw_q
        return torch.where(self._sel > 0, torch.minimum(torch.maximum(q, self._clo), self._chi), q).contiguous()

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        rnd = self.joint_random_range[1] > self.joint_random_range[0]
        m = None if mask is None else mask.bool()
        # first pose -> tip position -> targets (walk_v0.py:172-175, generate_targets :141-151); streams 19 / 20 / 21
        q1 = self._generate_qpos(mask, 19) if rnd else self._q0.expand(self.num_envs, -1).contiguous()
        E.reset(self.hm, self.state, mask, q1, None)
        E.reset_observation(self.hm, self.state, self._task, mask)
        nq, nv, n3 = self.cm.nq, self.cm.nv, 3 * self.ntip
        E.env_draw(self._draw_t, self._slo, self._shi, mask, self.episode, self._seed_u64, 20, env_index_base=self.env_index_base)
        tgt = self.obs[:, nq + nv:nq + nv + n3] + self._draw_t
        self.target_pos.copy_(tgt if m is None else torch.where(m[:, None], tgt, self.target_pos))
        # the episode starts from a second draw (walk_v0.py:181-184)
        q2 = self._generate_qpos(mask, 21) if rnd else q1
        E.reset(self.hm, self.state, mask, q2, self._v0.expand(self.num_envs, -1).contiguous())
        if m is None:
            self.episode += 1; self.step_count.zero_()
        else:
            self.episode += m.to(torch.int32); self.step_count.masked_fill_(m, 0)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}
