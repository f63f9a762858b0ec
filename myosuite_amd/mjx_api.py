"""MJX / MuJoCo-Playground style functional API on top of the fused HIP env-step.

Mirrors ``myosuite/envs/myo/mjx/mjx_base_env.py:63-86`` (``reset(rng) -> State``, ``step(State, action) -> State``) and the
pose task of ``playground_pose_v0.py:19-129`` (obs ``{"state": [qpos, qvel*sim_dt, act, target - qpos]}``, reward =
-angle_w*dist - ctrl_w*||act|| + bonus_w*bonus + penalty, ``done`` = dist > far_th, metrics ``*_reward`` / ``solved_frac``,
targets re-drawn and ``step_count`` zeroed in ``info`` on done / truncation).

Differences that follow from the batched device engine: one ``State`` holds all E environments (no ``jax.vmap``), its arrays
are torch tensors on the GPU, and ``step`` mutates the device buffers the State points to (the returned State is a new
object over the same storage -- there is no copy-on-write pytree).  ``rng`` is an integer Philox key.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict

import numpy as np
import torch

from . import engine as E
from .envs.pose_v0 import PoseEnvV0


@dataclasses.dataclass
class State:                                     # mujoco_playground.State fields
    data: E.BatchState
    obs: Dict[str, torch.Tensor]
    reward: torch.Tensor
    done: torch.Tensor
    metrics: Dict[str, torch.Tensor]
    info: Dict[str, Any]

    def replace(self, **kw) -> "State":
        return dataclasses.replace(self, **kw)


class MjxPoseEnv:
    """Batched counterpart of ``MjxPoseEnvV0`` (defaults: myo_registry / playground config of the MJX hand-pose task)."""

    def __init__(self, model: str = "hand", num_envs: int = 4096, target_jnt_range: dict = None, ctrl_dt: float = 0.02,
                 max_episode_steps: int = 100, norm_actions: bool = True, angle_reward_weight: float = 1.0,
                 ctrl_cost_weight: float = 1.0, bonus_weight: float = 4.0, pose_thd: float = 0.7,
                 far_th: float = 4 * np.pi / 2, device=None, seed: int = 0):
        from .envs import registry
        if target_jnt_range is None:
            target_jnt_range = registry.spec("myoHandPoseRandom-v0" if model == "hand" else "myoElbowPose1D6MRandom-v0")["kwargs"]["target_jnt_range"]
        from .model import synth
        sim_dt = synth.get_model(model).timestep
        self._env = PoseEnvV0(env_id=f"mjx-{model}-pose", model=model, num_envs=num_envs, device=device, seed=seed,
                              max_episode_steps=max_episode_steps, autoreset=False, target_jnt_range=target_jnt_range,
                              normalize_act=norm_actions, pose_thd=pose_thd, reset_type="random", target_type="generate",
                              frame_skip=int(round(ctrl_dt / sim_dt)),
                              weighted_reward_keys={"pose": angle_reward_weight, "act_reg": ctrl_cost_weight,
                                                    "bonus": bonus_weight, "penalty": 1.0})
        t = self._env._task
        t.obs_layout = 1; t.act_reg_mean = 0; t.obs_dt = sim_dt; t.far_th = float(far_th)
        self.max_episode_steps = max_episode_steps
        self.num_envs = num_envs
        self.cm = self._env.cm

    @property
    def observation_size(self) -> int:
        return self._env.obs_dim

    @property
    def action_size(self) -> int:
        return self.cm.nu

    def _metrics(self):
        r, w = self._env.rwd, self._env.rwd_keys_wt
        return {"pose_reward": w["pose"] * r[:, 0], "act_reg_reward": w["act_reg"] * r[:, 3], "bonus_reward": w["bonus"] * r[:, 1],
                "penalty_reward": r[:, 2], "solved_frac": r[:, 5] / self.max_episode_steps}

    def reset(self, rng: int) -> State:
        env = self._env
        env._task.obs_layout = 1
        E.pose_reset(env.hm, env.state, None, env._qlo, env._qhi, env._tlo, env._thi, env.target_jnt_value, env.episode,
                     env.step_count, int(rng), True, obs=env.obs, obs_layout=1)
        z = torch.zeros(self.num_envs, device=env.device)
        info = {"rng": int(rng), "target_angles": env.target_jnt_value, "step_count": env.step_count}
        return State(env.state, {"state": env.obs}, z, z.clone(), {k: z.clone() for k in
                     ("pose_reward", "act_reg_reward", "bonus_reward", "penalty_reward", "solved_frac")}, info)

    def step(self, state: State, action: torch.Tensor) -> State:
        env = self._env
        assert state.data is env.state, "State objects are views of this env's device buffers"
        a = torch.as_tensor(action, dtype=torch.float32, device=env.device).contiguous()
        E.env_step(env.hm, env.state, a, env._task)
        done = env.done.to(torch.float32)
        reward = env.rwd[:, 7].clone()
        metrics = {**state.metrics, **self._metrics()}
        # _get_info (playground_pose_v0.py:89-117): on done / truncation zero the step counter and re-draw the target
        need = (env.done | env.truncated)
        env.step_count.masked_fill_(need.bool(), 0)
        rng = state.info["rng"] + 1
        if bool(need.any()):
            new_t = torch.empty_like(env.target_jnt_value)
            E.uniform(new_t, seed=rng, stream_id=0x7A6)
            new_t = env._tlo + (env._thi - env._tlo) * new_t
            env.target_jnt_value.copy_(torch.where(need.bool()[:, None], new_t, env.target_jnt_value))
        info = {**state.info, "rng": rng, "step_count": env.step_count, "target_angles": env.target_jnt_value}
        return State(env.state, {"state": env.obs}, reward, done, metrics, info)


class MjxReachEnv:
    """Batched counterpart of ``MjxReachEnvV0`` (playground_reach_v0.py:11-165): obs ``{"state": [qpos, qvel*sim_dt, act, tip_pos,
    reach_err]}``, reward = -reach_weight*dist + bonus_scale*bonus - penalty_scale*(dist > far_th), ``done`` = dist > far_th
    (after two sim steps), targets re-drawn in ``info`` on done / truncation."""

    def __init__(self, num_envs: int = 4096, env_id: str = "myoHandReachRandom-v0", ctrl_dt: float = 0.02,
                 max_episode_steps: int = 100, norm_actions: bool = True, reach_weight: float = 1.0, bonus_scale: float = 4.0,
                 penalty_scale: float = 50.0, device=None, seed: int = 0):
        from .envs import registry
        from .envs.reach_v0 import ReachEnvV0
        kw = dict(registry.spec(env_id)["kwargs"])
        from .model import synth
        sim_dt = synth.get_model(kw["model"]).timestep
        kw.update(normalize_act=norm_actions, frame_skip=int(round(ctrl_dt / sim_dt)),
                  weighted_reward_keys={"reach": reach_weight, "bonus": bonus_scale, "penalty": penalty_scale})
        self._env = ReachEnvV0(env_id="mjx-" + env_id, num_envs=num_envs, device=device, seed=seed,
                               max_episode_steps=max_episode_steps, autoreset=False, **kw)
        t = self._env._task
        t.obs_layout = 1; t.obs_dt = sim_dt
        self.max_episode_steps = max_episode_steps
        self.num_envs = num_envs
        self.cm = self._env.cm

    @property
    def observation_size(self) -> int:
        return self._env.obs_dim

    @property
    def action_size(self) -> int:
        return self.cm.nu

    def _metrics(self):
        r, w = self._env.rwd, self._env.rwd_keys_wt
        return {"reach_reward": w["reach"] * r[:, 0], "bonus_reward": w["bonus"] * r[:, 1], "penalty_reward": w["penalty"] * r[:, 2],
                "solved_frac": r[:, 5] / self.max_episode_steps}

    def reset(self, rng: int) -> State:
        env = self._env
        env.reset(seed=int(rng))
        E.reset_observation(env.hm, env.state, env._task, None)        # first observation in the MJX layout
        z = torch.zeros(self.num_envs, device=env.device)
        info = {"rng": int(rng), "targets": env.target_pos, "step_count": env.step_count}
        return State(env.state, {"state": env.obs}, z, z.clone(),
                     {k: z.clone() for k in ("reach_reward", "bonus_reward", "penalty_reward", "solved_frac")}, info)

    def step(self, state: State, action: torch.Tensor) -> State:
        env = self._env
        assert state.data is env.state, "State objects are views of this env's device buffers"
        a = torch.as_tensor(action, dtype=torch.float32, device=env.device).contiguous()
        E.env_step(env.hm, env.state, a, env._task)
        done = env.done.to(torch.float32)
        reward = env.rwd[:, 7].clone()
        metrics = {**state.metrics, **self._metrics()}
        need = (env.done | env.truncated).bool()
        env.step_count.masked_fill_(need, 0)
        rng = state.info["rng"] + 1
        if bool(need.any()):
            u = torch.empty_like(env.target_pos)
            E.uniform(u, seed=rng, stream_id=0x7A7)
            new_t = env._tlo + (env._thi - env._tlo) * u
            env.target_pos.copy_(torch.where(need[:, None], new_t, env.target_pos))
        info = {**state.info, "rng": rng, "step_count": env.step_count, "targets": env.target_pos}
        return State(env.state, {"state": env.obs}, reward, done, metrics, info)
