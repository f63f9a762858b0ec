"""Batched reorient envs -- host-side mirror of myosuite/envs/myo/myobase/reorient_sar_v0.py
(``ProprioceptiveEnvV0`` :17-174, ``Geometries8EnvV0`` :177-262, ``Geometries100EnvV0`` :265-437).

obs keys ``hand_jnt, obj_pos, obj_vel, obj_rot, obj_des_rot, obj_err_pos, obj_err_rot, mlen, mvel, mforce`` (+ ``act``) = 200,
reward keys ``pos_align, rot_align, act_reg, drop, bonus``; every reset re-draws the object geometry and the desired
orientation (per-env model deltas: ``mm_state.geom_size_env``, ``mm_task.reor_axis_half / reor_des_rot``).

Object type ~ uniform over {capsule, ellipsoid, cylinder, box} and size ~ uniform over the type's table (2 rows each for
Reorient8, 25 each for Reorient100) exactly as the reference's reset (:388-406); collision of the non-capsule shapes
against the hand capsules uses the engine's segment-vs-convex narrow phase (one contact per pair).
"""
from __future__ import annotations

import collections
from typing import Optional

import numpy as np
import torch

from .. import engine as E
from ..model import synth
from .base_v0 import BaseV0
from .spaces import Box


class ReorientEnvV0(BaseV0):
    DEFAULT_OBS_KEYS = ["hand_jnt", "obj_pos", "obj_vel", "obj_rot", "obj_des_rot", "obj_err_pos", "obj_err_rot", "mlen",
                        "mvel", "mforce"]                                                    # reorient_sar_v0.py:97-108
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pos_align": 1.0, "rot_align": 1.0, "act_reg": 5.0, "drop": 5.0, "bonus": 10.0}   # :38-44

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None, max_episode_steps=50,
                 lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0, **kwargs):
        super().__init__(env_id, model, num_envs, device, seed, max_episode_steps, lanes_per_env, autoreset, env_index_base)
        self._setup(**kwargs)

    def _setup(self, geometries: str = "100", obs_keys=DEFAULT_OBS_KEYS, weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS,
               **kwargs):
        super()._setup(obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        gp = cm.arrays["GEOM_POS"].reshape(-1, 3).astype(np.float64)
        g = cm.names["geom"]
        self.pen_length = float(np.linalg.norm(gp[g["top"]] - gp[g["bot"]]))              # :84-91 (fixed at setup)
        self.tar_length = float(np.linalg.norm(gp[g["t_top"]] - gp[g["t_bot"]]))
        self.init_qpos = cm.qpos0.astype(np.float32).copy()
        self.init_qpos[:-6] *= 0; self.init_qpos[0] = -1.5                                # :113-114 palm up, hand open
        self._init_qpos_dev = torch.from_numpy(self.init_qpos).to(dev)
        self.size_tables_np = synth.reorient_tables(geometries)
        self._size_tables = torch.from_numpy(self.size_tables_np).to(dev).contiguous()
        self.geom_size = torch.zeros(n, 3, **f); self.axis_half = torch.zeros(n, **f); self.des_rot = torch.zeros(n, 3, **f)
        self.geom_type = torch.full((n,), 3, dtype=torch.int32, device=dev)
        self.state.set_geom_size_env(g["obj"], self.geom_size)
        self.state.set_geom_type_env(self.geom_type)
        self.obs_dim = (cm.nq - 6) + 3 + 6 + 3 + 3 + 3 + 3 + 3 * cm.nu + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_REORIENT), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_REORIENT)
        t.reor_obj_body = cm.body_id("Object"); t.reor_eps_site = cm.site_id("eps_ball"); t.reor_pen_length = self.pen_length
        t.reor_axis_half = self.axis_half.data_ptr(); t.reor_des_rot = self.des_rot.data_ptr()
        self._check_reward_keys(("pos_align", "rot_align", "act_reg", "drop", "bonus"))
        for i, k in enumerate(("pos_align", "rot_align", "act_reg", "drop", "bonus")):
            t.reor_w[i] = float(w.get(k, 0.0))
        t.reor_obs_muscle = 1
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _refresh_dicts(self):
        cm = self.cm
        o = self.obs
        sizes = [("hand_jnt", cm.nq - 6), ("obj_pos", 3), ("obj_vel", 6), ("obj_rot", 3), ("obj_des_rot", 3), ("obj_err_pos", 3),
                 ("obj_err_rot", 3)]
        if self._task.reor_obs_muscle:
            sizes += [("mlen", cm.nu), ("mvel", cm.nu), ("mforce", cm.nu)]
        sizes += [("act", cm.na)]
        od = collections.OrderedDict(time=self.state.time)
        k0 = 0
        for k, sz in sizes:
            od[k] = o[:, k0:k0 + sz]; k0 += sz
        od["obj_des_pos"] = od["obj_pos"] - od["obj_err_pos"]
        self.obs_dict = od
        r = self.rwd
        self.rwd_dict = collections.OrderedDict((k, r[:, i]) for i, k in enumerate(E.RWD_KEYS_REORIENT))
        self.rwd_dict["solved"] = self.rwd_dict["solved"] > 0.5
        self.rwd_dict["done"] = self.rwd_dict["done"] > 0.5

    def _rollout_fill_reset(self, ro):
        """the Geometries reset (object type / size / desired orientation draws, open-hand pose) runs inside the env-step launch
        where an env is a whole wavefront (mm_rollout; second, reset-observation pass for the envs the launch re-arms)"""
        if not (self.autoreset and self.hm.info(E.INFO_FOLDED_RESET) == 1 and not getattr(self, "fatigue_reset_random", False)):
            return
        ro.autoreset = 1
        ro.reor_init_qpos = self._init_qpos_dev.data_ptr()
        ro.reor_size_tables = self._size_tables.data_ptr(); ro.reor_ntab = int(self._size_tables.shape[1])
        ro.reor_tar_length = float(self.tar_length)
        ro.reor_geom_size_env = self.state.geom_size_env.data_ptr(); ro.reor_geom_type_env = self.state.geom_type_env.data_ptr()
        ro.reor_axis_half = self.axis_half.data_ptr(); ro.reor_des_rot = self.des_rot.data_ptr()
        ro.episode = self.episode.data_ptr(); ro.reset_seed = self._seed_u64
        if self.muscle_condition == "fatigue" and self.fatigue_reset_vec is not None:
            import numpy as np
            self._ro_fat_vec = torch.from_numpy(np.broadcast_to(np.asarray(self.fatigue_reset_vec, np.float32), (self.cm.na,)).copy()).to(self.device)
            ro.fat_reset_vec = self._ro_fat_vec.data_ptr()

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        E.reorient_reset_typed(self.hm, self.state, mask, self._init_qpos_dev, self._size_tables, self.axis_half,
                               self.des_rot, self.tar_length, self.episode, self.step_count, self._seed_u64)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}


class PenTwirlEnvV0(ReorientEnvV0):
    """Batched PenTwirl{Fixed,Random}EnvV0 -- mirror of myosuite/envs/myo/myobase/pen_v0.py:15-184.  Same observation /
    reward arithmetic as the SAR reorient env (which was derived from it) without the muscle length / velocity / force
    blocks: obs keys ``hand_jnt, obj_pos, obj_vel, obj_rot, obj_des_rot, obj_err_pos, obj_err_rot`` (+ ``act``) = 83; the pen
    geometry is fixed, ``random_target`` re-draws the desired orientation euler2quat([U(-1,1), U(-1,1), 0]) at reset."""
    DEFAULT_OBS_KEYS = ["hand_jnt", "obj_pos", "obj_vel", "obj_rot", "obj_des_rot", "obj_err_pos", "obj_err_rot"]   # pen_v0.py:16-26

    def _setup(self, random_target: bool = False, obs_keys=DEFAULT_OBS_KEYS,
               weighted_reward_keys=ReorientEnvV0.DEFAULT_RWD_KEYS_AND_WEIGHTS, **kwargs):
        BaseV0._setup(self, obs_keys=list(obs_keys), weighted_reward_keys=weighted_reward_keys, **kwargs)
        cm, n, dev = self.cm, self.num_envs, self.device
        f = dict(dtype=torch.float32, device=dev)
        sp = {nm: p for nm, _, p in [(k, 0, cm.arrays["SITE_POS"].reshape(-1, 3)[i]) for k, i in cm.names["site"].items()]}
        self.pen_length = float(np.linalg.norm(sp["object_top"] - sp["object_bottom"]))     # pen_v0.py:71-78
        self.tar_length = float(np.linalg.norm(sp["target_top"] - sp["target_bottom"]))
        self.random_target = bool(random_target)
        self.init_qpos = cm.qpos0.astype(np.float32).copy()
        self.init_qpos[:-6] *= 0; self.init_qpos[0] = -1.5                                  # pen_v0.py:85-86
        self._init_qpos_dev = torch.from_numpy(self.init_qpos).to(dev)
        self._axis_half = 0.5 * self.pen_length
        self.axis_half = torch.full((n,), self._axis_half, **f); self.des_rot = torch.zeros(n, 3, **f)
        self.obs_dim = (cm.nq - 6) + 3 + 6 + 3 + 3 + 3 + 3 + cm.na
        self.obs = torch.zeros(n, self.obs_dim, **f)
        self.rwd = torch.zeros(n, len(E.RWD_KEYS_REORIENT), **f)
        self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim),
                                     dtype=np.float32)
        w = self.rwd_keys_wt
        t = self._new_task(E.MM_TASK_REORIENT)
        t.reor_obj_body = cm.body_id("Object"); t.reor_eps_site = cm.site_id("eps_ball"); t.reor_pen_length = self.pen_length
        t.reor_axis_half = self.axis_half.data_ptr(); t.reor_des_rot = self.des_rot.data_ptr()
        self._check_reward_keys(("pos_align", "rot_align", "act_reg", "drop", "bonus"))
        for i, k in enumerate(("pos_align", "rot_align", "act_reg", "drop", "bonus")):
            t.reor_w[i] = float(w.get(k, 0.0))
        t.reor_obs_muscle = 0
        self._task = t
        self._seed_u64 = int(self.input_seed) if self.input_seed is not None else 0
        self.reset()

    def _rollout_fill_reset(self, ro):
        """pen-twirl keeps its separate reset call (mm_pen_reset: fixed geometry, other draws)"""

    def reset(self, seed=None, mask: Optional[torch.Tensor] = None, **kwargs):
        if seed is not None:
            self.seed(seed)
            self._seed_u64 = int(seed)
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        self._fatigue_reset(mask)
        rng = (-1.0, 1.0, -1.0, 1.0) if self.random_target else (0.0, 0.0, 0.0, 0.0)         # pen_v0.py:175-178
        E.pen_reset(self.hm, self.state, mask, self._init_qpos_dev, self._axis_half, rng, self.des_rot, self.tar_length,
                    self.episode, self.step_count, self._seed_u64)
        E.reset_observation(self.hm, self.state, self._task, mask)
        self._refresh_dicts()
        return self._obs_out(), {}
