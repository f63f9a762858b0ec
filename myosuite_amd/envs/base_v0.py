"""Batched counterpart of the reference's env runtime for musculoskeletal tasks.

Mirrors, for E environments at once and with tensors resident on the GPU:
  * ``MujocoEnv``   myosuite/envs/env_base.py:33   (step/forward/get_obs/reset, spaces, info dict,
                                                    get_env_state/set_env_state)
  * ``BaseV0``      myosuite/envs/myo/base_v0.py:14 (act appended to obs keys, muscle ctrl map,
                                                    sarcopenia / fatigue / reafferentation)
  * ``ObsVecDict``  myosuite/envs/obs_vec_dict.py:76-88 (ordered-key concat to float32)
The per-step arithmetic (ctrl map, fatigue, frame_skip x mj_step, final mj_forward, obs_dict,
reward_dict) is ONE fused HIP kernel launch (mm_env_step); this class only owns buffers and
bookkeeping.  API differences from the single-env reference: every returned array has a leading
[num_envs] dimension and is a torch tensor on the device.
"""
from __future__ import annotations

import collections
import collections.abc
from typing import Dict, Optional

import numpy as np
import torch

from .. import engine as E
from ..model import synth
from .spaces import Box

_MODEL_CACHE: Dict[tuple, object] = {}


def _weaken(spec):
    """sarcopenia (base_v0.py:60-67): actuator_gainprm[:, 2] *= 0.5 on EVERY actuator row (the peak force of a muscle; 0 on
    motors / servos, whose gain lives in gainprm[0]); biasprm is left untouched"""
    for a in spec.actuators:
        g = list(a.gainprm)
        if len(g) > 2:
            g[2] = 0.5 * g[2]
            a.gainprm = tuple(g)


def _compiled_model(name: str, muscle_condition: str):
    """Compiled model for ``model=``: the short name of a synthetic model (synth.builders()) or -- the reference's
    ``model_path`` (envs/env_base.py:72,96-106) -- the path of an MJCF file, imported by model/mjcf.py (includes into the
    ``simhive/myo_sim`` submodule resolve through $MYOSUITE_MYO_SIM_ROOT).  The sarcopenia edit is applied before compilation."""
    key = (name, muscle_condition == "sarcopenia")
    if key not in _MODEL_CACHE:
        edit = _weaken if muscle_condition == "sarcopenia" else None
        if isinstance(name, str) and name.lower().endswith(".xml"):
            from ..model import mjcf
            spec = mjcf.load(name)
            if edit is not None:
                edit(spec)
            cm = spec.compile()
            keys = getattr(spec, "keys", None)
            if keys:
                cm.key_qpos = np.array([k[0] for k in keys]); cm.key_qvel = np.array([k[1] for k in keys])
            _MODEL_CACHE[key] = cm
        else:
            _MODEL_CACHE[key] = synth.get_model(name) if edit is None else synth.compile_spec(name, edit)
    return _MODEL_CACHE[key]


class _LazyEnvState(collections.abc.Mapping):
    """info["state"] (env_base.py:614: get_env_state() of the step that just ended): materialised on first access -- five batched
    clones per env.step are not paid by callers that never look at it.  Read it before the next step() (as the reference's dict,
    it describes the state at the time it is taken; here that is the first access -- or freeze(), which step() calls ahead of an
    auto-reset so that info["state"] and info["obs_dict"] of one step describe the same state for finished envs too)."""

    def __init__(self, env):
        self._env, self._d = env, None

    def _get(self):
        if self._d is None:
            self._d = self._env.get_env_state()
        return self._d

    def freeze(self):
        """materialise now: step() calls this before an auto-reset rewrites the state rows of finished envs"""
        self._get()
        return self

    def __getitem__(self, k):
        return self._get()[k]

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())


class BaseV0:
    MYO_CREDIT = "MyoSuite: A contact-rich simulation suite for musculoskeletal motor control"

    def __init__(self, env_id: str, model: str, num_envs: int = 1, device=None, seed=None,
                 max_episode_steps: int = 0, lanes_per_env: int = 0, autoreset: bool = True, env_index_base: int = 0):
        self.env_id = env_id
        self.model_name = model
        self.num_envs = int(num_envs)
        self.max_episode_steps = int(max_episode_steps)
        self.autoreset = autoreset
        self.input_seed = seed
        self._lanes = lanes_per_env
        self._device = device
        self.env_index_base = int(env_index_base)   # global index of env 0 (env sharding over ranks): keys the Philox streams
        self.np_random = np.random.default_rng(seed)
        self.unwrapped = self

    # ------------------------------------------------------------------ setup
    def _setup(self, obs_keys, weighted_reward_keys, frame_skip=10, normalize_act=True, muscle_condition="",
               fatigue_reset_vec=None, fatigue_reset_random=False, reward_mode="dense", obs_range=(-10, 10),
               sites=None, precision="f32", fwd_carry=True, proprio_keys=None, visual_keys=None, **kwargs):
        """precision: "f32" (default) | "f64" | "f64_state" (or the MM_PREC_* value) -- the kernel family that steps the batch
        (include/myosim.h: fp64 arithmetic, optionally fp64 state rows; limit-rows-only models on Euler)"""
        if visual_keys:
            raise NotImplementedError("visual_keys need the renderer / visual encoders (env_base.py:222-300): out of this engine's scope")
        self.proprio_keys = None if proprio_keys is None else list(proprio_keys)       # env_base.py:112,557-576
        self.proprio_dict = {}
        self.muscle_condition = muscle_condition
        self.cm = _compiled_model(self.model_name, muscle_condition)
        self.precision = {"f32": E.MM_PREC_F32, "f64": E.MM_PREC_F64, "f64_state": E.MM_PREC_F64_STATE}.get(precision, precision)
        self.hm = E.HipModel(self.cm, lanes_per_env=self._lanes, device=self._device, precision=self.precision)
        self.device = self.hm.device
        # forward-pass carry (mm_task.fwd_carry): the trailing forward of env.step k hands its accelerations to the first substep
        # of env.step k + 1 (bit-identical, one pipeline pass in frame_skip + 1 saved) where the kernel family implements it
        self._fwd_carry = (torch.zeros(self.num_envs, 2 * self.cm.nv + 1, dtype=torch.float32, device=self.device)
                           if (fwd_carry and self.hm.info(E.INFO_FWD_CARRY) == 1) else None)
        cm = self.cm
        if cm.na > 0 and "act" not in obs_keys:       # base_v0.py:33-37
            obs_keys = list(obs_keys) + ["act"]
        self.obs_keys = list(obs_keys)
        # the fused launch writes the task's DEFAULT key order (ObsVecDict.obsdict2obsvec over the class's DEFAULT_OBS_KEYS + "act",
        # obs_vec_dict.py:76-88); a caller's own `obs_keys` (a subset / another order / "time": env_base.py:208-218 takes any key
        # of obs_dict) is served by step() / reset() as the concatenation of the obs_dict entries it names (_obs_out)
        self._kernel_obs_keys = list(type(self).DEFAULT_OBS_KEYS) + (["act"] if cm.na > 0 and "act" not in type(self).DEFAULT_OBS_KEYS else [])
        self._custom_obs = self.obs_keys != self._kernel_obs_keys
        self._custom_obs_ready = False
        self.rwd_keys_wt = dict(weighted_reward_keys)
        self.rwd_mode = reward_mode
        self.frame_skip = int(frame_skip)
        self.normalize_act = bool(normalize_act)
        self.fatigue_reset_vec = fatigue_reset_vec
        self.fatigue_reset_random = fatigue_reset_random
        self.tip_sids, self.target_sids = [], []
        if sites:
            for s in sites:
                self.tip_sids.append(cm.site_id(s))
                self.target_sids.append(cm.site_id(s + "_target"))
        n = self.num_envs
        dev = self.device
        self.state = E.BatchState(self.hm, n)
        self.state.env_index_base = self.env_index_base
        self.step_count = torch.zeros(n, dtype=torch.int32, device=dev)
        self.episode = torch.zeros(n, dtype=torch.int32, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.truncated = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.last_ctrl = torch.zeros(n, cm.nu, dtype=torch.float32, device=dev)
        # action space (env_base.py:143-155)
        if self.normalize_act:
            lo, hi = -np.ones(cm.nu), np.ones(cm.nu)
        else:
            cr = cm.arrays["ACT_CTRLRANGE"].reshape(-1, 2)
            lo, hi = cr[:, 0].copy(), cr[:, 1].copy()
        self.action_space = Box(lo, hi, dtype=np.float32, seed=self.input_seed)
        # muscle conditions (base_v0.py:60-79)
        self.fat_MA = self.fat_MR = self.fat_MF = None
        self.reaf = (-1, -1)
        if muscle_condition == "fatigue":
            f = dict(dtype=torch.float32, device=dev)
            self.fat_MA = torch.zeros(n, cm.na, **f)
            self.fat_MR = torch.ones(n, cm.na, **f)
            self.fat_MF = torch.zeros(n, cm.na, **f)
            self._fat_draws = None     # random fatigue reset (fatigue.py:84-90): two [n, na] Philox draws, allocated on first use
        elif muscle_condition == "reafferentation":
            self.reaf = (cm.names["actuator"]["EIP"], cm.names["actuator"]["EPL"])
        self.init_qpos = cm.qpos0.astype(np.float32).copy()
        self.init_qvel = np.zeros(cm.nv, np.float32)
        self.obs_dict: Dict[str, torch.Tensor] = {}
        self.rwd_dict: Dict[str, torch.Tensor] = {}
        self._obs_range = obs_range

    # ------------------------------------------------------------------ properties
    @property
    def dt(self) -> float:                    # env_base.py:660-662
        return self.cm.timestep * self.frame_skip

    @property
    def horizon(self) -> int:
        return self.max_episode_steps

    @property
    def id(self) -> str:
        return self.env_id

    @property
    def time(self) -> torch.Tensor:
        return self.state.time

    # ------------------------------------------------------------------ what the reference's own env test touches (tests/test_envs.py:54-128)
    # mjModel field name -> section of the compiled model (read-only numpy copies; per-env deltas live in `state`)
    _MJMODEL_ARRAYS = {"body_mass": "BODY_MASS", "body_pos": "BODY_POS", "body_quat": "BODY_QUAT", "body_inertia": "BODY_INERTIA",
                       "body_parentid": "BODY_PARENT", "jnt_range": "JNT_RANGE", "jnt_type": "JNT_TYPE", "jnt_qposadr": "JNT_QPOSADR",
                       "jnt_dofadr": "JNT_DOFADR", "jnt_bodyid": "JNT_BODYID", "jnt_stiffness": "JNT_STIFFNESS", "dof_damping": "DOF_DAMPING",
                       "dof_armature": "DOF_ARMATURE", "dof_frictionloss": "DOF_FRICTIONLOSS", "geom_size": "GEOM_SIZE", "geom_type": "GEOM_TYPE",
                       "geom_pos": "GEOM_POS", "geom_bodyid": "GEOM_BODYID", "site_pos": "SITE_POS", "site_bodyid": "SITE_BODYID",
                       "tendon_lengthspring": "TENDON_LENGTHSPRING", "tendon_stiffness": "TENDON_STIFFNESS", "tendon_range": "TENDON_RANGE",
                       "actuator_lengthrange": "ACT_LENGTHRANGE", "actuator_gainprm": "ACT_GAINPRM", "actuator_biasprm": "ACT_BIASPRM",
                       "actuator_dynprm": "ACT_DYNPRM", "actuator_ctrlrange": "ACT_CTRLRANGE", "actuator_forcerange": "ACT_FORCERANGE",
                       "actuator_gear": "ACT_GEAR", "actuator_acc0": "ACT_ACC0", "actuator_trnid": "ACT_TRNID", "qpos0": "QPOS0",
                       "qpos_spring": "QPOS_SPRING"}

    @property
    def mj_model(self):
        """read-only view of the compiled model under mjModel's names: the dimensions (nq, nv, nu, na, nbody, njnt, ngeom, nsite,
        ntendon), `opt.timestep`, the `<kind>_names` lists the reference's wrapper offers (`env.mj_model.actuator_names.index("ECRL")`,
        agents/baseline_Reflex/ReflexCtrInterface.py:274) and the constant arrays scripts read (`body_mass`, `jnt_range`,
        `actuator_lengthrange`, `actuator_gainprm` / `biasprm`, `tendon_lengthspring`, ...: numpy COPIES of the compiled sections --
        writing to them changes nothing; per-env model edits go through the env's kwargs / `state.set_*_env`).  Not a MuJoCo object:
        there is none."""
        import types
        v = getattr(self, "_mj_model_view", None)
        if v is None:
            cm = self.cm
            v = types.SimpleNamespace(nq=cm.nq, nv=cm.nv, nu=cm.nu, na=cm.na, nbody=cm.nbody, njnt=cm.njnt, ngeom=cm.ngeom, nsite=cm.nsite,
                                      ntendon=cm.ntendon, opt=types.SimpleNamespace(timestep=float(cm.arrays["OPT_F"][0])), names=cm.names)
            for kind, d in cm.names.items():
                setattr(v, f"{kind}_names", [n for n, _ in sorted(d.items(), key=lambda kv: kv[1])])
            for field, sec in self._MJMODEL_ARRAYS.items():
                if sec in cm.arrays:
                    setattr(v, field, np.array(cm.arrays[sec]))
            self._mj_model_view = v
        return v

    @property
    def mj_data(self):
        """view of the batched state with mjData's field names (time, qpos, qvel, act, ctrl, qacc_warmstart): tensors [num_envs, ...],
        plus the name-addressed getters of the reference's wrapper for single-dof joints (`get_joint_qpos(name)` / `get_joint_qvel`:
        a [num_envs] tensor)."""
        import types
        s, cm = self.state, self.cm

        def _jadr(name, key):
            return int(cm.arrays[key][cm.joint_id(name)])
        return types.SimpleNamespace(time=s.time, qpos=s.qpos, qvel=s.qvel, act=s.act, ctrl=self.last_ctrl, qacc_warmstart=s.qacc_warmstart,
                                     get_joint_qpos=lambda name: s.qpos[:, _jadr(name, "JNT_QPOSADR")],
                                     get_joint_qvel=lambda name: s.qvel[:, _jadr(name, "JNT_DOFADR")])

    def get_obs_dict(self, *sim_args, state=None):
        """the reference's `get_obs_dict(sim)` / `get_obs_dict(mj_model, mj_data)` on the env's OWN simulation: the obs_dict of the
        current state, which the launch computed (tasks with a torch restatement -- Pose -- also take another `state`)"""
        if state is not None:
            raise NotImplementedError(f"{type(self).__name__}.get_obs_dict on a foreign state: only the current obs_dict is available")
        return self.obs_dict

    def get_reward_dict(self, obs_dict):
        if obs_dict is not self.obs_dict:
            raise NotImplementedError(f"{type(self).__name__}.get_reward_dict on a foreign obs_dict: only the current rwd_dict is available")
        return self.rwd_dict

    def get_exteroception(self, **kwargs) -> dict:
        """env_base.py:578-582 = get_visuals: no visual keys can be configured here (rendering is out of scope), so nothing to return"""
        return {}

    def __reduce__(self):
        """pickle / copy as the reference's EzPickle does: by re-running the constructor with the arguments of `registry.make`"""
        from . import registry
        args = getattr(self, "_make_args", None)
        if args is None:
            raise TypeError(f"{type(self).__name__} was not built by registry.make: cannot be pickled")
        return (registry._remake, (args,))

    def seed(self, seed=None):
        self.input_seed = seed
        self.np_random = np.random.default_rng(seed)
        return [seed]

    def get_input_seed(self):
        return self.input_seed

    # ------------------------------------------------------------------ fatigue (fatigue.py:82-99)
    def _fatigue_reset(self, mask: Optional[torch.Tensor]):
        if self.muscle_condition != "fatigue":
            return
        n, na = self.num_envs, self.cm.na
        if self.fatigue_reset_random:
            # fatigue.py:84-90: non_fatigued, active_percentage ~ U[0,1)^na.  Drawn on the device from Philox streams 22 / 23 of
            # (seed, GLOBAL env index, episode) -- like every other reset draw, so a shard of envs (env_index_base) reproduces its
            # slice of the unsharded batch and two ranks with one seed never start from the same fatigue state
            assert self.fatigue_reset_vec is None, "Cannot use 'fatigue_reset_vec' if fatigue_reset_random=True."
            if self._fat_draws is None:
                f = dict(dtype=torch.float32, device=self.device)
                self._fat_draws = (torch.zeros(n, na, **f), torch.zeros(n, na, **f), torch.zeros(na, **f), torch.ones(na, **f))
            nf, ap, lo, hi = self._fat_draws
            seed = int(getattr(self, "_seed_u64", self.input_seed if self.input_seed is not None else 0))
            E.env_draw(nf, lo, hi, mask, self.episode, seed, 22, env_index_base=self.env_index_base)
            E.env_draw(ap, lo, hi, mask, self.episode, seed, 23, env_index_base=self.env_index_base)
            MA, MR, MF = nf * ap, nf * (1 - ap), 1 - nf
        else:
            # deterministic reset (to rest, or to fatigue_reset_vec): one launch, no temporaries
            # (the attribute is read on every reset, as the reference does -- fatigue.py:82-99, base_v0.py:124 -- so a vector
            # assigned later, e.g. through MyoVecEnv.set_attr, takes effect; the device copy is rebuilt only when it changed)
            vec = None
            if self.fatigue_reset_vec is not None:
                src = np.broadcast_to(np.asarray(self.fatigue_reset_vec, np.float32), (na,)).copy()
                cached = getattr(self, "_fat_vec_src", None)
                if cached is None or not np.array_equal(cached, src):
                    self._fat_vec_src = src
                    self._fat_vec = torch.from_numpy(src).to(self.device)
                vec = self._fat_vec
            E.fatigue_reset(self.fat_MA, self.fat_MR, self.fat_MF, mask, vec)
            return
        if mask is None:
            self.fat_MA.copy_(MA); self.fat_MR.copy_(MR); self.fat_MF.copy_(MF)
        else:
            m = mask.bool()[:, None]
            self.fat_MA.copy_(torch.where(m, MA, self.fat_MA))
            self.fat_MR.copy_(torch.where(m, MR, self.fat_MR))
            self.fat_MF.copy_(torch.where(m, MF, self.fat_MF))

    def set_fatigue_reset_random(self, fatigue_reset_random):
        self.fatigue_reset_random = fatigue_reset_random

    # ------------------------------------------------------------------ state get/set (env_base.py:688-759)
    def get_env_state(self) -> dict:
        s = self.state
        return dict(time=s.time.clone(), qpos=s.qpos.clone(), qvel=s.qvel.clone(),
                    act=s.act.clone() if self.cm.na > 0 else None, qacc_warmstart=s.qacc_warmstart.clone(),
                    step_count=self.step_count.clone())

    def set_env_state(self, state_dict: dict):
        s = self.state
        s.time.copy_(state_dict["time"]); s.qpos.copy_(state_dict["qpos"]); s.qvel.copy_(state_dict["qvel"])
        if self.cm.na > 0 and state_dict.get("act") is not None:
            s.act.copy_(state_dict["act"])
        if "qacc_warmstart" in state_dict:
            s.qacc_warmstart.copy_(state_dict["qacc_warmstart"])
        if "step_count" in state_dict:
            self.step_count.copy_(state_dict["step_count"])

    def forward(self):
        """`env.sim.forward()` + `get_obs()` of the reference after a state edit (env_base.py:434-459, 720-760): one forward pass of
        the CURRENT state of every env (no stepping, no counters) refreshing `obs`, `obs_dict`, `rwd_dict`."""
        E.reset_observation(self.hm, self.state, self._task, None)
        self._refresh_dicts()

    def get_obs(self, update_proprioception=True, update_exteroception=False):
        """env_base.py:434-459: the observation vector of the current state (recomputed: valid after `set_env_state` or a direct
        edit of `state.qpos` / `mj_data.qpos`)."""
        self.forward()
        if update_proprioception and self.proprio_keys is not None:
            self.proprio_dict = self.get_proprioception()[2]
        return self._obs_out()

    def evaluate_success(self, paths, logger=None, successful_steps=5):
        """env_base.py:798-824, same arithmetic: a path counts as solved when its `env_infos["solved"]` sums above
        `successful_steps`; returns the percentage (and logs rwd_sparse / rwd_dense / success_percentage when given a logger)."""
        num_paths = len(paths)
        num_success = sum(1 for p in paths if np.sum(np.asarray(p["env_infos"]["solved"]) * 1.0) > successful_steps)
        success_percentage = num_success * 100.0 / num_paths
        if logger:
            logger.log_kv("rwd_sparse", np.mean([np.mean(p["env_infos"]["rwd_sparse"]) for p in paths]))
            logger.log_kv("rwd_dense", np.mean([np.sum(p["env_infos"]["rwd_dense"]) / self.horizon for p in paths]))
            logger.log_kv("success_percentage", success_percentage)
        return success_percentage

    def get_proprioception(self, obs_dict=None):
        """env_base.py:557-576: (time, proprio vector, proprio dict) over `proprio_keys`, or (None, None, None) when none are configured"""
        if self.proprio_keys is None:
            return None, None, None
        od = self.obs_dict if obs_dict is None else obs_dict
        n = self.num_envs
        pd = collections.OrderedDict(time=od["time"])
        for k in self.proprio_keys:
            pd[k] = od[k]
        vec = torch.cat([od[k].reshape(n, -1).to(torch.float32) for k in self.proprio_keys], dim=1) if self.proprio_keys else \
            torch.zeros(n, 0, device=self.device)
        return pd["time"], vec, pd

    # ------------------------------------------------------------------ info dict (env_base.py:585-616)
    def get_env_infos(self) -> dict:
        if self.proprio_keys is not None:
            self.proprio_dict = self.get_proprioception()[2]
        return collections.OrderedDict(
            time=self.obs_dict["time"], rwd_dense=self.rwd_dict["dense"], rwd_sparse=self.rwd_dict["sparse"],
            solved=self.rwd_dict["solved"], done=self.rwd_dict["done"], obs_dict=self.obs_dict, visual_dict={},
            proprio_dict=self.proprio_dict, rwd_dict=self.rwd_dict, state=_LazyEnvState(self))

    def _new_task(self, task_id: int, do_forward: bool = True) -> "E.mm_task":
        """mm_task with the fields every task shares (frame_skip, ctrl map, fatigue state, output buffers, counters).
        Call after self.obs / self.rwd exist."""
        t = E.mm_task()
        t.task = task_id; t.nsubsteps = self.frame_skip; t.normalize_act = int(self.normalize_act)
        t.do_forward = int(do_forward); t.fatigue = int(self.muscle_condition == "fatigue")
        t.max_episode_steps = self.max_episode_steps
        if self.fat_MA is not None:
            t.fat_MA, t.fat_MR, t.fat_MF = self.fat_MA.data_ptr(), self.fat_MR.data_ptr(), self.fat_MF.data_ptr()
        t.fat_F, t.fat_R, t.fat_r = 0.00912, 0.1 * 0.00094, 10 * 15          # fatigue.py:9-11
        t.obs = self.obs.data_ptr(); t.obs_dim = self.obs_dim; t.rwd = self.rwd.data_ptr()
        t.done = self.done.data_ptr(); t.truncated = self.truncated.data_ptr()
        t.step_count = self.step_count.data_ptr(); t.ctrl_out = self.last_ctrl.data_ptr()
        t.fwd_carry = self._fwd_carry.data_ptr() if self._fwd_carry is not None else None
        t.reaf_src, t.reaf_dst = self.reaf
        t.obs_dt = self.dt
        return t

    # ------------------------------------------------------------------ step (env_base.py:377-407, base_v0.py:82-118)
    def _check_reward_keys(self, supported):
        """weighted_reward_keys may re-weight or drop the task's reward terms; a key the fused launch does not sum (the reference
        would add ANY key of rwd_dict, env_base.py:440-446) is refused instead of being silently left out of `dense`"""
        unknown = [k for k, wt in self.rwd_keys_wt.items() if k not in supported and float(wt) != 0.0]
        if unknown:
            raise NotImplementedError(f"weighted_reward_keys {unknown}: the launch sums {list(supported)} for this task")

    def _obs_out(self) -> torch.Tensor:
        """the observation vector step() / reset() hand out: the kernel's buffer for the task's default keys, else the caller's
        `obs_keys` gathered from obs_dict (as the reference's obsdict2obsvec does over ITS obs_keys)"""
        if not self._custom_obs:
            return self.obs
        n = self.num_envs
        missing = [k for k in self.obs_keys if k not in self.obs_dict]
        if missing:
            raise KeyError(f"obs_keys {missing} are not keys of this task's obs_dict {list(self.obs_dict)}")
        out = torch.cat([self.obs_dict[k].reshape(n, -1).to(torch.float32) for k in self.obs_keys], dim=1)
        if not self._custom_obs_ready:
            self.kernel_obs_dim, self.obs_dim = self.obs_dim, int(out.shape[1])
            self.observation_space = Box(self._obs_range[0] * np.ones(self.obs_dim), self._obs_range[1] * np.ones(self.obs_dim), dtype=np.float32)
            self._custom_obs_ready = True
        return out

    def step(self, a, **kwargs):
        """One fused kernel launch: ctrl map / fatigue, frame_skip x mj_step, final forward, task obs + reward; then the
        masked auto-reset (always enqueued, no host sync; a no-op for envs that continue).  Subclasses provide
        ``_task``, ``_refresh_dicts()`` and ``reset(mask=...)``."""
        a = torch.as_tensor(a, dtype=torch.float32, device=self.device)
        if a.dim() == 1:
            a = a.expand(self.num_envs, -1)
        a = a.contiguous()
        E.env_step(self.hm, self.state, a, self._task)
        self._refresh_dicts()
        # fresh arrays, as the reference returns: the buffers behind rwd_dict / obs_dict are rewritten by the next step
        reward = (self.rwd_dict["dense"] if self.rwd_mode == "dense" else self.rwd_dict["sparse"]).clone()
        terminated = self.done.bool()
        truncated = self.truncated.bool() & ~terminated
        info = self.get_env_infos()
        obs = self._obs_out()
        if self.autoreset:
            # info describes the step that just ended (its obs_dict views would otherwise show the post-reset observation of
            # finished envs): snapshot before the masked reset rewrites self.obs
            info["final_obs"] = obs.clone()
            info["obs_dict"] = collections.OrderedDict((k, v.clone()) for k, v in self.obs_dict.items())
            info["state"].freeze()      # get_env_state() of the step that just ended (env_base.py:614), not of the re-armed episode
            self.reset(mask=(self.done | self.truncated))
            obs = self._obs_out()
        return obs, reward, terminated, truncated, info

    # ------------------------------------------------------------------ rollout step (one launch)
    def rollout_setup(self, ep_stats: Optional[torch.Tensor] = None, action_seed: int = 0, action_out: Optional[torch.Tensor] = None):
        """Prepare `rollout_step`: per-env (return, length, solved) accumulators and, for the Pose family, the masked
        auto-reset folded into the env-step launch (mm_rollout).  Tasks without a folded reset re-arm finished envs with
        their reset call, driven by the reset mask the launch writes."""
        n, dev = self.num_envs, self.device
        self._ro_stats = ep_stats if ep_stats is not None else torch.zeros(n, 3, dtype=torch.float32, device=dev)
        self._ro_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._ro_action_out = action_out
        ro = E.mm_rollout()
        ro.action_seed = int(action_seed)
        ro.ep_stats = self._ro_stats.data_ptr(); ro.reset_mask = self._ro_mask.data_ptr()
        ro.action_out = action_out.data_ptr() if action_out is not None else None
        ro.autoreset = 0
        self._rollout_fill_reset(ro)
        self._ro = ro
        self._ro_sig = self._rollout_signature()
        return self._ro_stats

    def _rollout_signature(self):
        """what the folded reset fields of mm_rollout were filled from: a later reset(seed=...) / seed() / assignment of
        reset_type or fatigue_reset_vec (a NEW vector object: in-place edits of the old one are not seen) re-fills them at the
        next rollout_step, so the folded and the separate reset path cannot drift apart"""
        return (getattr(self, "_seed_u64", 0), getattr(self, "reset_type", None), getattr(self, "target_type", None), id(self.fatigue_reset_vec),
                bool(self.fatigue_reset_random), bool(self.autoreset))

    def _rollout_fill_reset(self, ro):
        """tasks whose reset is folded into the launch fill mm_rollout's reset fields here (PoseEnvV0)"""

    def rollout_step(self, action: Optional[torch.Tensor] = None, stream_id: int = 0, events=None):
        """env.step + auto-reset + episode statistics for rollout harnesses, without the gym-level dict plumbing: ONE kernel
        launch for the Pose family (action ~ U[0,1) drawn in the kernel when `action` is None, benchmarks/mjx_benchmark.py:29),
        one more (the task's masked reset) for the others.  Returns (obs, reward_row_view, reset_mask): obs holds the first
        observation of the new episode for re-armed envs; views are rewritten by the next call."""
        if self._custom_obs:
            raise NotImplementedError("rollout_step / the on-device PPO read the kernel's observation buffer (the task's default obs_keys); "
                                      "custom obs_keys are served by step() / reset()")
        ro = self._ro
        sig = self._rollout_signature()
        if sig != self._ro_sig:
            ro.autoreset = 0
            self._rollout_fill_reset(ro)
            self._ro_sig = sig
        if action is not None:
            assert action.shape == (self.num_envs, self.cm.nu) and action.dtype == torch.float32 and action.is_contiguous()
            ro.action = action.data_ptr()
        else:
            ro.action = None
        ro.action_stream = int(stream_id)
        if events is not None:      # (start, end) torch.cuda.Event pair around the fused env-step launch alone
            events[0].record()
        E.rollout_step(self.hm, self.state, self._task, ro)
        if events is not None:
            events[1].record()
        if not ro.autoreset and self.autoreset:
            self.reset(mask=self._ro_mask)
        return self.obs, self.rwd, self._ro_mask

    def capture_step_graph(self, warmup: int = 2):
        """Capture ``step()`` (fused env-step launch + masked auto-reset + the few tensor ops between them) into a HIP graph.

        Returns ``(graph, action, outputs)``: fill ``action`` ([num_envs, nu], in place), call ``graph.replay()``; ``outputs`` is
        the ``(obs, reward, terminated, truncated, info)`` tuple of the captured call, whose tensors are rewritten by every
        replay.  One replay costs one launch instead of ~10, which is what bounds small models (elbow: 0.2 ms per eager step).
        The warm-up steps advance the envs (call ``reset()`` afterwards if that matters).  Every reset draw is a device-side
        Philox stream keyed on the device-resident episode counters, so replays keep drawing fresh values."""
        action = torch.zeros(self.num_envs, self.cm.nu, dtype=torch.float32, device=self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):                      # first launches set kernel attributes: keep them out of the capture
            for _ in range(max(1, warmup)):
                self.step(action)
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outputs = self.step(action)
        return graph, action, outputs

    def close(self):
        pass
