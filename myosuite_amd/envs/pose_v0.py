"""Batched PoseEnvV0 -- host-side mirror of myosuite/envs/myo/myobase/pose_v0.py:15-257.

Same constructor kwargs, obs keys (``qpos, qvel, pose_err`` + ``act``), reward keys and
weights, reset/target types.  One ``step()`` = one fused HIP launch (mm_env_step) for all
``num_envs`` environments (+ one masked reset launch when auto-reset is on).
"""
from __future__ import annotations

import collections
import math
from typing import Optional

import numpy as np
import torch

from .. import engine as E
from .base_v0 import BaseV0
from .spaces import Box


class PoseEnvV0(BaseV0):
    FAR_TH = 4 * math.pi / 2                                               # pose_v0.py:118
    DEFAULT_OBS_KEYS = ["qpos", "qvel", "pose_err"]                        # pose_v0.py:17
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pose": 1.0, "bonus": 4.0, "act_reg": 1.0, "penalty": 50}   # pose_v0.py:18-23
    SWITCH_POSES = ((-0.145125, 0.92524251, 1.08978337, 1.39425813, -0.78286243, -0.77179383, -0.15042819, 0.64445902),
                    (-0.12756566, 0.06741454, 1.51352705, 0.91777418, -0.63884237, 0.22452487, 0.42103326, 0.4139465))   # pose_v0.py:201-228

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=100,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, viz_site_targets: tuple = None, target_jnt_range: dict = None, target_jnt_value=None,
               reset_type="init", target_type="generate", obs_keys=DEFAULT_OBS_KEYS,
               weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS, pose_thd=0.35, weight_bodyname=None,
               weight_range=None, do_forward: bool = True, **kwargs):
        self.reset_type = reset_type
        self.target_type = target_type
        self.pose_thd = float(pose_thd)
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, sites=viz_site_targets,
                       **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        # resolve joint demands (pose_v0.py:63-75)
        if target_jnt_range:
            ids = [cm.joint_id(j) for j in target_jnt_range]
            assert ids == list(range(cm.nq)), "target_jnt_range must list every joint in qpos order"
            self.target_jnt_range = np.array([target_jnt_range[j] for j in target_jnt_range], np.float32)
            tv = np.mean(self.target_jnt_range, axis=1)
        else:
            self.target_jnt_range = None
            tv = np.asarray(target_jnt_value, np.float32)
            assert tv.shape == (cm.nq,)
        self.target_jnt_value = torch.from_numpy(np.tile(tv.astype(np.float32), (n, 1))).to(dev).contiguous()
        trange = self.target_jnt_range if self.target_jnt_range is not None else np.stack([tv, tv], 1)
        if self.target_type == "fixed":
            trange = np.stack([tv, tv], 1) if self.target_jnt_range is None else self.target_jnt_range
        self._tlo = torch.from_numpy(np.ascontiguousarray(trange[:, 0], np.float32)).to(dev)
        self._thi = torch.from_numpy(np.ascontiguousarray(trange[:, 1], np.float32)).to(dev)
        jr = cm.jnt_range.astype(np.float32)
        self._qlo = torch.from_numpy(np.ascontiguousarray(jr[:, 0])).to(dev)
        self._qhi = torch.from_numpy(np.ascontiguousarray(jr[:, 1])).to(dev)
        # outputs
        self.obs_dim = cm.nq + cm.nv + cm.nq + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_POSE), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        # per-episode weight of one body (pose_v0.py:177-187): a per-env model delta (mm_state.body_mass_env)
        self.weight_bodyname, self.weight_range = weight_bodyname, weight_range
        if weight_bodyname is not None:
            self.body_mass = torch.full((n,), float(cm.arrays["BODY_MASS"][cm.names["body"][weight_bodyname]]), **f)
            self.state.set_body_mass_env(cm.names["body"][weight_bodyname], self.body_mass)
            self._wlo = torch.tensor([float(weight_range[0])], **f); self._whi = torch.tensor([float(weight_range[1])], **f)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_POSE, do_forward)
        t.pose_thd = self.pose_thd; t.far_th = self.FAR_TH
        self._check_reward_keys(("pose", "bonus", "act_reg", "penalty"))
        t.w_pose = float(w.get("pose", 0.0)); t.w_bonus = float(w.get("bonus", 0.0))
        t.w_act_reg = float(w.get("act_reg", 0.0)); t.w_penalty = float(w.get("penalty", 0.0))
        t.target_jnt_value = self.target_jnt_value.data_ptr()
        t.obs_layout = 0; t.act_reg_mean = 1
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    # ------------------------------------------------------------------ obs / reward dicts
    def _refresh_dicts(self):
        cm = self.cm
        nq, nv, na = cm.nq, cm.nv, cm.na
        o = self.obs
        self.obs_dict = collections.OrderedDict(
            time=self.state.time, qpos=o[:, :nq], qvel=o[:, nq:nq + nv], pose_err=o[:, nq + nv:2 * nq + nv],
            act=o[:, 2 * nq + nv:2 * nq + nv + na])
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_POSE))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def get_obs_dict(self, *sim_args, state=None):
        """pose_v0.py:100-111 on the current (or given) batched state, as torch ops."""
        s = state if state is not None else self.state
        d = collections.OrderedDict()
        d["time"] = s.time
        d["qpos"] = s.qpos.clone()
        d["qvel"] = s.qvel * self.dt
        d["act"] = s.act.clone() if self.cm.na > 0 else torch.zeros_like(s.qpos)
        d["pose_err"] = self.target_jnt_value - d["qpos"]
        return d

    def get_reward_dict(self, obs_dict):
        """pose_v0.py:113-140 (vectorised over the leading env dimension)."""
        pose_dist = torch.linalg.norm(obs_dict["pose_err"], dim=-1)
        act_mag = torch.linalg.norm(obs_dict["act"], dim=-1)
        if self.cm.na != 0:
            act_mag = act_mag / self.cm.na
        far_th = getattr(self, "FAR_TH", PoseEnvV0.FAR_TH)
        rwd = collections.OrderedDict((
            ("pose", -1.0 * pose_dist),
            ("bonus", 1.0 * (pose_dist < self.pose_thd) + 1.0 * (pose_dist < 1.5 * self.pose_thd)),
            ("penalty", -1.0 * (pose_dist > far_th)),
            ("act_reg", -1.0 * act_mag),
            ("sparse", -1.0 * pose_dist),
            ("solved", pose_dist < self.pose_thd),
            ("done", pose_dist > far_th)))
        rwd["dense"] = sum(wt * rwd[k] for k, wt in self.rwd_keys_wt.items())
        return rwd

    def _rollout_fill_reset(self, ro):
        """the Pose family's reset (targets ~ U(target range), qpos ~ U(joint range) or qpos0) runs inside the env-step launch;
        per-episode model deltas (carried weight) and the "none" / "switch" modes keep the separate reset call"""
        if (self.weight_bodyname is None and self.reset_type in ("init", "random") and self.target_type in ("generate", "fixed")
                and self.muscle_condition != "fatigue" and self.autoreset):
            generate = self.target_type == "generate"
            self._ro_tfix = None if generate else self.target_jnt_value[0].clone().contiguous()
            ro.autoreset = 1
            ro.random_qpos = int(self.reset_type == "random")
            ro.qlo, ro.qhi = self._qlo.data_ptr(), self._qhi.data_ptr()
            ro.tlo = self._tlo.data_ptr() if generate else self._ro_tfix.data_ptr()
            ro.thi = self._thi.data_ptr() if generate else self._ro_tfix.data_ptr()
            ro.target = self.target_jnt_value.data_ptr(); ro.episode = self.episode.data_ptr()
            ro.reset_seed = self._seed_u64

    def _torch_obs(self):
        """Observation vector of the current state as torch ops on the state rows (Pose needs no forward pass)."""
        d = self.get_obs_dict()
        return torch.cat([d[k].reshape(self.num_envs, -1) for k in self._kernel_obs_keys], dim=1).to(torch.float32)

    # ------------------------------------------------------------------ reset (pose_v0.py:174-257)
    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, reset_qpos=None, reset_qvel=None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        if self.reset_type in (None, "none"):
            # no state reset; only targets (and counters) are refreshed.  The fatigue state is not reset either
            # (pose_v0.py: "NOTE: fatigue is also not reset in this case": BaseV0.reset is skipped)
            keep = self.get_env_state()
        else:
            self._fatigue_reset(mask)
        if self.weight_bodyname is not None:     # weight ~ U(weight_range), Philox stream 16 of (seed, env, episode)
            E.env_draw(self.body_mass, self._wlo, self._whi, mask, self.episode, self._seed_u64, 16,
                       env_index_base=self.env_index_base)
        if self.target_type not in ("generate", "fixed", "switch"):
            raise TypeError("Unknown Target type: {}".format(self.target_type))          # pose_v0.py:150-151
        if getattr(self, "_ro", None) is not None:
            self._ro.reset_seed = self._seed_u64
        generate = self.target_type == "generate"
        random_q = self.reset_type == "random" and reset_qpos is None
        switch_prev = self.target_jnt_value.clone() if self.target_type == "switch" else None
        # targets ~ U(target range) and (optionally) qpos ~ U(joint range): Philox keyed by (seed, env, episode)
        E.pose_reset(self.hm, self.state, mask, self._qlo, self._qhi,
                     self._tlo if generate else self.target_jnt_value[0].contiguous(),
                     self._thi if generate else self.target_jnt_value[0].contiguous(),
                     self.target_jnt_value, self.episode, self.step_count, self._seed_u64, random_q,
                     obs=self.obs, obs_layout=0)
        if switch_prev is not None:
            # pose_v0.py:197-243: alternate between the reference's two hard-coded 8-joint poses (per env, at its resets)
            if self.cm.nq != 8:
                raise NotImplementedError("target_type='switch' alternates between two hard-coded 8-joint poses in the "
                                          "reference (pose_v0.py:201-239); this model has nq = %d" % self.cm.nq)
            A = torch.tensor(self.SWITCH_POSES[0], dtype=torch.float32, device=self.device)
            B = torch.tensor(self.SWITCH_POSES[1], dtype=torch.float32, device=self.device)
            new_t = torch.where((switch_prev[:, :1] != A[0]), A[None, :], B[None, :])
            sel = torch.ones(self.num_envs, 1, dtype=torch.bool, device=self.device) if mask is None else mask.bool()[:, None]
            self.target_jnt_value.copy_(torch.where(sel, new_t, switch_prev))
        simple = reset_qpos is None and self.reset_type not in (None, "none") and switch_prev is None
        if reset_qpos is not None:
            q = torch.as_tensor(reset_qpos, dtype=torch.float32, device=self.device).expand(self.num_envs, -1).contiguous()
            v = None if reset_qvel is None else torch.as_tensor(reset_qvel, dtype=torch.float32, device=self.device).expand(self.num_envs, -1).contiguous()
            E.reset(self.hm, self.state, mask, q, v)
        if self.reset_type in (None, "none"):
            m = None if mask is None else mask.bool()
            for k in ("time", "qpos", "qvel", "act", "qacc_warmstart"):
                if keep[k] is None:
                    continue
                dst = getattr(self.state, k)
                dst.copy_(keep[k] if m is None else torch.where(m.view(-1, *([1] * (dst.dim() - 1))), keep[k], dst))
        if not simple:   # state was overwritten after the kernel wrote its observation
            self.obs.copy_(self._torch_obs()) if mask is None else self.obs.copy_(
                torch.where(mask.bool()[:, None], self._torch_obs(), self.obs))
        self._refresh_dicts()
        return self._obs_out(), {}
