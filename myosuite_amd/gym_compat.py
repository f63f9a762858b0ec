"""gym / gymnasium / Stable-Baselines3 boundary: the doors the reference's agents come through.

    import myosuite_amd.gym_compat as mg
    mg.register_all()                                  # what `import myosuite` does (myosuite/__init__.py:25-67)
    env = gym.make("myoHandPoseRandom-v0")             # single env, numpy in / out, TimeLimit from the registry horizon
    venv = gym.make_vec("myoHandPoseRandom-v0", num_envs=4096, vectorization_mode="vector_entry_point")   # gymnasium >= 0.29
    venv = mg.make_vec_env("myoHandPoseRandom-v0", n_envs=4096)   # stable_baselines3.common.env_util.make_vec_env's signature
                                                                   # (agents/sb3_job_script.py:49): ONE batched env, not n copies

Everything behind these doors is the batched engine: a `SingleEnv` is a 1-env batch, a `MyoVecEnv` is an n-env batch whose
step is one fused kernel launch.  `gymnasium` / `stable_baselines3` are optional: when importable, ids are registered with
gymnasium, spaces are gymnasium spaces and the classes derive from gymnasium.Env / SB3's VecEnv (so VecNormalize & co. accept
them); when absent (this image), the same classes work on the numpy shim in envs/spaces.py.
"""
from __future__ import annotations

import importlib
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .envs import registry
from .envs import spaces as shim


def _gym():
    for name in ("gymnasium", "gym"):
        try:
            return importlib.import_module(name)
        except ImportError:
            continue
    return None


def _box(low, high, seed=None):
    g = _gym()
    if g is not None and hasattr(g, "spaces"):
        return g.spaces.Box(low=np.asarray(low, np.float32), high=np.asarray(high, np.float32), dtype=np.float32)
    return shim.Box(low, high, dtype=np.float32, seed=seed)


_EnvBase = object
_g = _gym()
if _g is not None and hasattr(_g, "Env"):
    _EnvBase = _g.Env
try:
    from stable_baselines3.common.vec_env import VecEnv as _VecBase      # noqa: F401
    _HAVE_SB3 = True
except ImportError:
    _VecBase = object
    _HAVE_SB3 = False


def _rebuild_single(env_id, device, seed, kwargs):
    return SingleEnv(env_id, device=device, seed=seed, **kwargs)


class SingleEnv(_EnvBase):
    """`gym.make(id)`: one environment with the reference's single-env signatures (envs/env_base.py:395-407, 640-654):
    reset(seed=...) -> (obs, {}), step(a) -> (obs, reward, terminated, truncated=False, info); numpy float32 observations.
    Episode truncation is gym.make's TimeLimit wrapper's job (max_episode_steps of the registry)."""
    metadata: Dict[str, Any] = {"render_modes": []}

    def __init__(self, env_id: str, device=None, seed=None, **kwargs):
        self._ctor = dict(env_id=env_id, device=None if device is None else str(device), seed=seed, kwargs=dict(kwargs))
        kwargs = dict(kwargs)
        kwargs.pop("max_episode_steps", None)
        self._env = registry.make(env_id, num_envs=1, device=device, seed=seed, autoreset=False, max_episode_steps=0, **kwargs)
        b = self._env
        self.observation_space = _box(b.observation_space.low, b.observation_space.high)
        self.action_space = _box(b.action_space.low, b.action_space.high, seed=seed)
        self.spec = None

    def __reduce__(self):          # pickle.loads(pickle.dumps(env)) as the reference's EzPickle envs (tests/test_envs.py:80): rebuilt by the constructor
        c = self._ctor
        return (_rebuild_single, (c["env_id"], c["device"], c["seed"], c["kwargs"]))

    @property
    def unwrapped(self):
        return self

    def __getattr__(self, name):         # mj-model-free attributes the reference's tests touch: obs_dict, rwd_dict, dt, ...
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._env, name)

    @staticmethod
    def _np0(d):
        """env 0 of a dict of batched tensors as numpy (the single-env door speaks numpy, like the reference)"""
        return {k: (v[0].detach().cpu().numpy() if hasattr(v, "detach") else v) for k, v in d.items()}

    def reset(self, *, seed=None, options=None, **kwargs):
        obs, info = self._env.reset(seed=seed, **kwargs)
        return obs[0].cpu().numpy(), info

    def step(self, a):
        obs, rwd, term, trunc, info = self._env.step(np.asarray(a, np.float32)[None])
        st = info.get("state")
        flat = {"time": float(info["time"][0]), "rwd_dense": float(info["rwd_dense"][0]), "rwd_sparse": float(info["rwd_sparse"][0]),
                "solved": bool(info["solved"][0]), "done": bool(info["done"][0]), "obs_dict": self._np0(info["obs_dict"]),
                "rwd_dict": self._np0(info["rwd_dict"]), "visual_dict": {}, "proprio_dict": self._np0(info.get("proprio_dict", {})),
                "state": None if st is None else self._np0({k: v for k, v in st.items() if v is not None})}   # env_base.py:604-615
        return obs[0].cpu().numpy(), float(rwd[0]), bool(term[0]), False, flat

    def get_proprioception(self, obs_dict=None):
        t, vec, d = self._env.get_proprioception()
        if d is None:
            return None, None, None
        return float(t[0]), vec[0].cpu().numpy(), self._np0(d)

    def get_obs_dict(self, *sim_args, **kw):
        return self._np0(self._env.get_obs_dict())

    def get_obs(self, update_proprioception=True, update_exteroception=False):      # env_base.py:434-459
        return self._env.get_obs(update_proprioception, update_exteroception)[0].cpu().numpy()

    def get_env_state(self):                                                         # env_base.py:688-718
        return self._np0({k: v for k, v in self._env.get_env_state().items() if v is not None})

    def set_env_state(self, state_dict):                                             # env_base.py:720-760
        dev = self._env.device
        self._env.set_env_state({k: torch.as_tensor(np.asarray(v), device=dev)[None] for k, v in state_dict.items() if v is not None})

    def get_reward_dict(self, obs_dict=None):
        return self._np0(self._env.get_reward_dict(self._env.obs_dict))

    def render(self):
        return None

    def close(self):
        self._env.close()


class MyoVecEnv(_VecBase):
    """n environments as ONE batched env behind Stable-Baselines3's VecEnv protocol (numpy in / out, auto-reset with
    ``terminal_observation`` / ``TimeLimit.truncated`` in the per-env info dicts, step_async / step_wait, get_attr / set_attr /
    env_method / env_is_wrapped / seed).  Also offers gymnasium.vector.VectorEnv's 5-tuple through `step5`."""

    def __init__(self, env_id: str, n_envs: int, seed=None, device=None, **env_kwargs):
        self.env = registry.make(env_id, num_envs=n_envs, device=device, seed=seed, autoreset=True, **env_kwargs)
        b = self.env
        self.num_envs = int(n_envs)
        self.observation_space = _box(b.observation_space.low, b.observation_space.high)
        self.action_space = _box(b.action_space.low, b.action_space.high, seed=seed)
        if _HAVE_SB3:
            _VecBase.__init__(self, self.num_envs, self.observation_space, self.action_space)
        self.single_observation_space, self.single_action_space = self.observation_space, self.action_space
        self._actions = None
        self.render_mode = None

    # ---- SB3 VecEnv
    def reset(self):
        obs, _ = self.env.reset()
        return obs.cpu().numpy()

    def step_async(self, actions):
        self._actions = np.asarray(actions, np.float32).reshape(self.num_envs, -1)

    def step_wait(self):
        obs, rwd, term, trunc, info = self.env.step(self._actions)
        done = (term | trunc).cpu().numpy()
        tr = trunc.cpu().numpy()
        final = info["final_obs"].cpu().numpy() if done.any() else None
        solved = info["solved"].cpu().numpy()
        infos: List[Dict[str, Any]] = [{"solved": bool(solved[i])} for i in range(self.num_envs)]
        for i in np.nonzero(done)[0]:
            infos[i]["terminal_observation"] = final[i]
            infos[i]["TimeLimit.truncated"] = bool(tr[i])
        return obs.cpu().numpy(), rwd.cpu().numpy().astype(np.float32), done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def seed(self, seed: Optional[int] = None):
        self.env.seed(seed)
        return [seed] * self.num_envs

    def _idx(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name: str, indices=None):
        v = getattr(self.env, attr_name)
        return [v for _ in self._idx(indices)]

    def set_attr(self, attr_name: str, value, indices=None):
        setattr(self.env, attr_name, value)

    def env_method(self, method_name: str, *method_args, indices=None, **method_kwargs):
        r = getattr(self.env, method_name)(*method_args, **method_kwargs)
        return [r for _ in self._idx(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._idx(indices)]

    def get_images(self) -> Sequence[Optional[np.ndarray]]:
        return [None] * self.num_envs

    def render(self, mode: Optional[str] = None):
        return None

    @property
    def unwrapped(self):
        return self

    # ---- host buffers in, host buffers out, nothing else: the boundary a CPU-side learner sees
    def step_host(self, actions):
        """``actions`` [n, nu] float32 on the host -> ``(obs, reward, done)`` numpy arrays on the host, at the cost of the transfers
        and nothing more: one pinned H2D copy, the fused env-step launch (episode statistics + auto-reset inside / right behind it,
        `BaseV0.rollout_step`), three pinned D2H copies, ONE stream synchronisation; no per-env Python objects.  ``obs`` holds the
        first observation of the new episode for envs whose ``done`` flag (terminated | truncated) is set -- the SB3 convention
        without ``terminal_observation``.  The returned arrays are views of pinned staging buffers, rewritten by the next call.
        This is the path the PCIe-inclusive rate in DESIGN.md is measured on (benchmarks/host_boundary.py)."""
        b = self.env
        if getattr(self, "_host", None) is None:
            if getattr(b, "_ro", None) is None:
                b.rollout_setup()
            pin = lambda t: torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._host = dict(a_dev=torch.empty(self.num_envs, b.cm.nu, dtype=torch.float32, device=b.device),
                              a=torch.empty(self.num_envs, b.cm.nu, dtype=torch.float32, pin_memory=True),
                              obs=pin(b.obs), rwd=pin(b.rwd), done=pin(b._ro_mask),
                              col=list(b.rwd_dict).index("dense" if b.rwd_mode == "dense" else "sparse"))
        h = self._host
        h["a"].numpy()[...] = np.asarray(actions, np.float32).reshape(self.num_envs, -1)
        h["a_dev"].copy_(h["a"], non_blocking=True)
        obs, rwd, mask = b.rollout_step(h["a_dev"])
        h["obs"].copy_(obs, non_blocking=True); h["rwd"].copy_(rwd, non_blocking=True); h["done"].copy_(mask, non_blocking=True)
        torch.cuda.current_stream(b.device).synchronize()
        return h["obs"].numpy(), h["rwd"].numpy()[:, h["col"]], h["done"].numpy().view(np.bool_)

    # ---- gymnasium.vector.VectorEnv flavour
    def reset5(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed)
        return obs.cpu().numpy(), info

    def step5(self, actions):
        obs, rwd, term, trunc, info = self.env.step(np.asarray(actions, np.float32).reshape(self.num_envs, -1))
        return obs.cpu().numpy(), rwd.cpu().numpy(), term.cpu().numpy(), trunc.cpu().numpy(), info


def make_vec_env(env_id: str, n_envs: int = 1, seed: Optional[int] = None, start_index: int = 0, monitor_dir=None,
                 wrapper_class=None, env_kwargs: Optional[dict] = None, vec_env_cls=None, vec_env_kwargs=None,
                 monitor_kwargs=None, wrapper_kwargs=None) -> MyoVecEnv:
    """Drop-in for ``stable_baselines3.common.env_util.make_vec_env`` (agents/sb3_job_script.py:49,52): same signature, but the
    n environments are one batched env on the GPU (the per-env Monitor / wrapper_class / vec_env_cls arguments have nothing to
    wrap and must be left at their defaults)."""
    if wrapper_class is not None or vec_env_cls is not None or monitor_dir is not None:
        raise ValueError("myosuite_amd.gym_compat.make_vec_env builds ONE batched env: per-env wrappers / monitors / vec_env_cls do not apply")
    return MyoVecEnv(env_id, n_envs, seed=seed, **(env_kwargs or {}))


def _vector_entry_point(env_id: str):
    def make(num_envs: int = 1, **kwargs):
        kwargs.pop("max_episode_steps", None)
        return MyoVecEnv(env_id, num_envs, **kwargs)
    return make


def _single_entry_point(env_id: str):
    def make(**kwargs):
        return SingleEnv(env_id, **kwargs)
    return make


_REGISTERED: List[str] = []


def register_all(force: bool = False) -> List[str]:
    """Register every id of envs/registry.py with gymnasium (or gym) when it is importable -- the side effect `import myosuite`
    has in the reference (myosuite/__init__.py:25-67; envs/env_variants.py for the Sarc / Fati / Reaf variants).  Returns the
    ids registered; an empty list (and no error) when neither package is installed."""
    g = _gym()
    if g is None or not hasattr(g, "register"):
        return []
    existing = set()
    try:
        existing = set(g.envs.registry.keys()) if hasattr(g.envs.registry, "keys") else set(g.envs.registry.env_specs.keys())
    except Exception:
        pass
    out = []
    for env_id, sp in registry.registry_specs().items():
        if env_id in existing and not force:
            continue
        kw = dict(id=env_id, entry_point=_single_entry_point(env_id), max_episode_steps=sp["max_episode_steps"])
        try:
            g.register(vector_entry_point=_vector_entry_point(env_id), **kw)     # gymnasium >= 0.29
        except TypeError:
            g.register(**kw)
        out.append(env_id)
    _REGISTERED.extend(out)
    return out
