"""Batched TorsoEnvV0 -- host-side mirror of myosuite/envs/myo/myobase/torso_v0.py:15-148.

The reference class is PoseEnvV0 with three differences, all kept: ``far_th = pi`` (torso_v0.py:119), a ``done`` weight in the
default reward keys (:23; weight 0) and a target that is fixed at the mean of ``target_jnt_range`` -- ``reset()`` never
re-draws it (:135-148).  ``pose_err = target - qpos[:18]`` (:114): the synthetic myoTorso has exactly the 18 listed joints.
"""
from __future__ import annotations

import math

from .pose_v0 import PoseEnvV0


class TorsoEnvV0(PoseEnvV0):
    DEFAULT_OBS_KEYS = ["qpos", "qvel", "pose_err"]                                                     # torso_v0.py:17
    DEFAULT_RWD_KEYS_AND_WEIGHTS = {"pose": 1.0, "bonus": 4.0, "act_reg": 1.0, "penalty": 50, "done": 0}   # torso_v0.py:18-24
    FAR_TH = math.pi                                                                                    # torso_v0.py:119

    def _setup(self, pose_thd=0.25, reset_type="init", weighted_reward_keys=DEFAULT_RWD_KEYS_AND_WEIGHTS, **kwargs):
        kwargs.pop("target_type", None)
        super()._setup(pose_thd=pose_thd, reset_type=reset_type, target_type="fixed",
                       weighted_reward_keys=weighted_reward_keys, **kwargs)
