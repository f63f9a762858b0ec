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


MJX_ITERATIONS = MJX_LS_ITERATIONS = 6


def _mjx_solver_options(env):
    """MjxMyoBase.preprocess_spec (envs/myo/mjx/mjx_base_env.py:50-51) overwrites the loaded model's solver budget:
    `spec.option.iterations = 6; spec.option.ls_iterations = 6`.  The same on this env's model handle (mm_model_set_option)."""
    env.hm.set_option("iterations", MJX_ITERATIONS)
    env.hm.set_option("ls_iterations", MJX_LS_ITERATIONS)


class MjxPoseEnv:
    """Batched counterpart of ``MjxPoseEnvV0`` (defaults: myo_registry / playground config of the MJX hand-pose task)."""

    def __init__(self, model: str = "hand", num_envs: int = 4096, target_jnt_range: dict = None, ctrl_dt: float = 0.02,
                 max_episode_steps: int = 100, norm_actions: bool = True, angle_reward_weight: float = 1.0,
                 ctrl_cost_weight: float = 1.0, bonus_weight: float = 4.0, pose_thd: float = 0.7,
                 far_th: float = 4 * np.pi / 2, device=None, seed: int = 0, **variant_kw):
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
                                                    "bonus": bonus_weight, "penalty": 1.0}, **variant_kw)
        _mjx_solver_options(self._env)
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
                "penalty_reward": r[:, 2].clone(), "solved_frac": r[:, 5] / self.max_episode_steps}

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
        new_t = torch.empty_like(env.target_jnt_value)         # no host sync: draw for everyone, keep where not needed
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
                 penalty_scale: float = 50.0, device=None, seed: int = 0, **variant_kw):
        from .envs import registry
        from .envs.reach_v0 import ReachEnvV0
        kw = dict(registry.spec(env_id)["kwargs"])
        kw.update(variant_kw)
        from .model import synth
        sim_dt = synth.get_model(kw["model"]).timestep
        kw.update(normalize_act=norm_actions, frame_skip=int(round(ctrl_dt / sim_dt)),
                  weighted_reward_keys={"reach": reach_weight, "bonus": bonus_scale, "penalty": penalty_scale})
        self._env = ReachEnvV0(env_id="mjx-" + env_id, num_envs=num_envs, device=device, seed=seed,
                               max_episode_steps=max_episode_steps, autoreset=False, **kw)
        _mjx_solver_options(self._env)
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
        u = torch.empty_like(env.target_pos)                   # no host sync: draw for everyone, keep where not needed
        E.uniform(u, seed=rng, stream_id=0x7A7)
        new_t = env._tlo + (env._thi - env._tlo) * u
        env.target_pos.copy_(torch.where(need[:, None], new_t, env.target_pos))
        info = {**state.info, "rng": rng, "step_count": env.step_count, "targets": env.target_pos}
        return State(env.state, {"state": env.obs}, reward, done, metrics, info)



# ---------------------------------------------------------------------------------------------------------------------
# muscle-condition variants of the MJX ids (envs/myo/mjx/myo_registry.py:13, 54-98)
MJX_VARIANT_PREFIXES = ("MjxSarc", "MjxFati", "MjxReaf")
ALLOWED_FATIGUE_OBS_KEYS = ("MA", "MR", "MF")            # fatigue_jax.py:13
_BASE_ENVS = tuple(f"Mjx{task}{kind}-v0" for task in ("ElbowPose", "FingerPose", "HandReach") for kind in ("Fixed", "Random"))
# register_environment_with_variants registers the base id and its `MjxFati` twin (the Sarc / Reaf twins are commented out there)
ALL_ENVS = tuple(n for b in _BASE_ENVS for n in (b, b[:3] + "Fati" + b[3:]))


def get_base_env_name(env_name: str) -> str:
    """myo_registry.get_base_env_name (myo_registry.py:92-98): the id without its muscle-condition prefix."""
    return env_name[:3] + env_name[7:] if env_name[:7] in MJX_VARIANT_PREFIXES else env_name


class MjxFatigueEnv:
    """Batched counterpart of ``FatigueWrapper`` (fatigue_jax.py:176-304) around an Mjx env: every action goes through
    ``1 / (1 + exp(-5 (a - 0.5)))`` (the wrapper switches the inner env's own normalisation off and applies it itself, :232), the
    3CC-r model turns the muscles' target loads into the active fraction MA, which is what the simulation receives as control (:241-245).
    Here all of that is inside the fused launch (``mm_task.fatigue``; the kernel's ``compute_act`` is checked against the
    reference's ``fatigue.py`` vectors).  The reference keeps MA / MR / MF in ``state.data.userdata``; here they are the three
    ``[E, na]`` device tensors of ``fatigue_state(state)`` (also ``state.info["fatigue_state"]``).  ``fatigue_obs_keys`` appends
    them to ``obs["state"]`` in the order MA, MR, MF (:280-292); ``fatigue_reset_vec`` / ``fatigue_reset_random`` as :203-209."""

    def __init__(self, env, fatigue_reset_vec=None, fatigue_reset_random: bool = False, fatigue_obs_keys=()):
        assert all(k in ALLOWED_FATIGUE_OBS_KEYS for k in fatigue_obs_keys), \
            f"Invalid fatigue_obs_keys: {list(fatigue_obs_keys)}. Allowed keys are: {list(ALLOWED_FATIGUE_OBS_KEYS)}"
        self.env = env
        inner = env._env
        assert inner.muscle_condition == "fatigue" and inner._task.fatigue == 1 and inner._task.normalize_act == 1
        inner.fatigue_reset_vec = fatigue_reset_vec
        inner.fatigue_reset_random = bool(fatigue_reset_random)
        self.fatigue_obs_keys = list(fatigue_obs_keys)
        self.num_envs, self.cm, self.max_episode_steps = env.num_envs, env.cm, env.max_episode_steps
        self._env = inner                                  # (TrainingWrapper reaches the BaseV0 env through `_env`)

    @property
    def observation_size(self) -> int:
        return self.env.observation_size + self.cm.na * len(self.fatigue_obs_keys)

    @property
    def action_size(self) -> int:
        return self.env.action_size

    def fatigue_state(self, state: State = None) -> Dict[str, torch.Tensor]:
        inner = self._env
        return {"MA": inner.fat_MA, "MR": inner.fat_MR, "MF": inner.fat_MF}

    def set_fatigue_reset_random(self, fatigue_reset_random):      # fatigue_jax.py:294-295
        self._env.fatigue_reset_random = bool(fatigue_reset_random)

    def _with_fatigue(self, st: State) -> State:
        fs = self.fatigue_state()
        obs = st.obs
        if self.fatigue_obs_keys and "state" in obs:
            obs = {**obs, "state": torch.cat([obs["state"]] + [fs[k] for k in ALLOWED_FATIGUE_OBS_KEYS if k in self.fatigue_obs_keys], dim=-1)}
        return st.replace(obs=obs, info={**st.info, "fatigue_state": fs})

    def reset(self, rng: int) -> State:
        st = self.env.reset(rng)
        inner = self._env
        inner._seed_u64 = int(rng) & 0xFFFFFFFFFFFFFFFF        # the random fatigue reset is keyed like the other reset draws
        inner._fatigue_reset(None)
        return self._with_fatigue(st)

    def step(self, state: State, action: torch.Tensor) -> State:
        return self._with_fatigue(self.env.step(state, action))

# ---------------------------------------------------------------------------------------------------------------------
# make() of myosuite/envs/myo/mjx/__init__.py:109-199 and the training wrapper of mujoco_playground / brax
PPO_CONFIG = dict(                                   # myosuite/envs/myo/mjx/__init__.py:43-67
    num_timesteps=50_000_000, learning_rate=3e-4, discounting=0.97, gae_lambda=0.95, entropy_cost=0.001, clipping_epsilon=0.3,
    max_grad_norm=1.0, action_repeat=1, num_minibatches=32, num_updates_per_batch=8, batch_size=256, unroll_length=10,
    reward_scaling=1.0, normalize_observations=True, num_evals=16, num_eval_envs=128, num_resets_per_eval=1,
    network_factory=dict(policy_hidden_layer_sizes=(64, 64, 64), value_hidden_layer_sizes=(64, 64, 64),
                         policy_obs_key="state", value_obs_key="state"))


def get_default_config(env_name: str) -> dict:
    base = dict(ctrl_dt=0.02, sim_dt=0.002, num_envs=4096, max_episode_steps=100, impl="hip", norm_actions=True)
    if "Reach" in env_name:
        base.update(reward_config=dict(reach_weight=1.0, bonus_scale=4.0, penalty_scale=50.0), far_th=0.35)
    else:
        base.update(reward_config=dict(angle_reward_weight=1.0, ctrl_cost_weight=1.0, pose_thd=0.35, far_th=4 * np.pi / 2,
                                       bonus_weight=4.0))
    return base


def make(env_name: str, config_overrides: dict = None, num_envs: int = None, device=None, seed: int = 0):
    """``myosuite.envs.myo.mjx.make`` for the ids the reference registers there: Mjx{Elbow,Finger}Pose{Fixed,Random}-v0,
    MjxHandReach{Fixed,Random}-v0 and the ``MjxFati`` twin of each (myo_registry.py:54-81: base env class wrapped in
    ``FatigueWrapper``; its ``fatigue_reset_vec / fatigue_reset_random / fatigue_obs_keys`` come out of ``config_overrides``,
    fatigue_jax.py:297-310).  ``impl`` is always this engine."""
    from .envs import registry
    if env_name not in ALL_ENVS:
        raise KeyError(f"unknown MJX env {env_name!r}. Available envs: {list(ALL_ENVS)}")
    variant = env_name[:7] if env_name[:7] in MJX_VARIANT_PREFIXES else ""
    base_name = get_base_env_name(env_name)
    ov = dict(config_overrides or {})
    fat_cfg = dict(fatigue_reset_vec=None, fatigue_reset_random=False, fatigue_obs_keys=())       # FatigueWrapper.DEFAULT_MUSCLE_CONFIG
    vkw = {}
    if variant == "MjxFati":
        fat_cfg.update(ov.pop("fatigue_config", {}) or {})
        for k in list(fat_cfg):
            if k in ov:
                fat_cfg[k] = ov.pop(k)
        vkw = dict(muscle_condition="fatigue")
    cfg = get_default_config(base_name)
    cfg.update(ov)
    if variant == "MjxFati":
        cfg["norm_actions"] = True       # the wrapper disables the env's normalisation and applies the same map to every action itself
    n = int(num_envs if num_envs is not None else cfg["num_envs"])
    fixed = "Fixed" in base_name
    rc = cfg["reward_config"]
    if base_name.startswith("MjxElbowPose") or base_name.startswith("MjxFingerPose"):
        elbow = base_name.startswith("MjxElbowPose")
        ref_id = ("myoElbowPose1D6M" if elbow else "myoFingerPose") + ("Fixed" if fixed else "Random") + "-v0"
        env = MjxPoseEnv(model="elbow" if elbow else "finger", num_envs=n, target_jnt_range=registry.spec(ref_id)["kwargs"]["target_jnt_range"],
                         ctrl_dt=cfg["ctrl_dt"], max_episode_steps=cfg["max_episode_steps"], norm_actions=cfg["norm_actions"],
                         angle_reward_weight=rc["angle_reward_weight"], ctrl_cost_weight=rc["ctrl_cost_weight"],
                         bonus_weight=rc["bonus_weight"], pose_thd=rc["pose_thd"], far_th=rc["far_th"], device=device, seed=seed, **vkw)
    else:
        env = MjxReachEnv(num_envs=n, env_id="myoHandReach" + ("Fixed" if fixed else "Random") + "-v0", ctrl_dt=cfg["ctrl_dt"],
                          max_episode_steps=cfg["max_episode_steps"], norm_actions=cfg["norm_actions"],
                          reach_weight=rc["reach_weight"], bonus_scale=rc["bonus_scale"], penalty_scale=rc["penalty_scale"],
                          device=device, seed=seed, **vkw)
    if variant == "MjxFati":
        env = MjxFatigueEnv(env, **fat_cfg)
    return env


class TrainingWrapper:
    """What ``mujoco_playground.wrapper.wrap_for_brax_training`` adds for PPO (benchmarks/mjx_benchmark_PPO.py:59): episode
    truncation and auto-reset.  Brax's AutoResetWrapper restores the FIRST state of the run on done; here the finished envs
    are re-drawn from the env's own reset distribution (one masked reset launch), which is the better-mixed equivalent."""

    def __init__(self, env):
        self.env = env
        self.num_envs = env.num_envs
        self.observation_size, self.action_size = env.observation_size, env.action_size

    # ---- the rollout facade the on-device learner drives (myosuite_amd/ppo.py): env-step + auto-reset + episode statistics as ONE
    # launch where the task's reset is folded into it (Pose), the launch + the masked reset + the reset observation otherwise
    @property
    def _inner(self):
        return self.env._env

    device = property(lambda self: self._inner.device)
    obs_dim = property(lambda self: self._inner.obs_dim)
    cm = property(lambda self: self._inner.cm)
    obs = property(lambda self: self._inner.obs)
    rwd = property(lambda self: self._inner.rwd)
    truncated = property(lambda self: self._inner.truncated)

    def rollout_setup(self, **kw):
        if getattr(self.env, "fatigue_obs_keys", None):
            raise NotImplementedError("the fused rollout writes the task's observation row; fatigue_obs_keys are served by reset() / step()")
        self._inner.autoreset = True
        self._inner._task.obs_layout = 1
        return self._inner.rollout_setup(**kw)

    def rollout_step(self, action):
        inner = self._inner
        obs, rw, mask = inner.rollout_step(action)
        if not inner._ro.autoreset:        # the separate reset wrote the env's default layout: first observation in the MJX layout
            E.reset_observation(inner.hm, inner.state, inner._task, mask)
        return inner.obs, rw, mask

    def reset(self, rng: int) -> State:
        return self.env.reset(rng)

    def step(self, state: State, action: torch.Tensor) -> State:
        st = self.env.step(state, action)
        inner = self.env._env
        need = (inner.done | inner.truncated)
        done = need.to(torch.float32)
        truncation = (inner.truncated.bool() & ~inner.done.bool()).to(torch.float32)
        inner.reset(mask=need)                              # masked reset launch; obs of the reset envs in the env's layout
        E.reset_observation(inner.hm, inner.state, inner._task, need.to(torch.uint8).contiguous())
        info = {**st.info, "truncation": truncation}
        out = State(st.data, {"state": inner.obs}, st.reward, done, st.metrics, info)
        return self.env._with_fatigue(out) if hasattr(self.env, "_with_fatigue") else out    # (fatigue_obs_keys of an MjxFati env)
