"""Env-id registry mirroring ``myosuite/envs/myo/myobase/__init__.py``.

Same ids, same kwargs (targets, thresholds, reset/target types, horizons) and the
same muscle-condition variants (``myoSarc*``, ``myoFati*``, ``myoReaf*``;
myobase/__init__.py:17-49).  ``model_path`` is replaced by the name of a
synthetic model (the real MJCF is an empty submodule in the reference).

    env = make("myoHandPoseRandom-v0", num_envs=4096)     # batched, tensors on the GPU
"""
from __future__ import annotations

import copy
from typing import Callable, Dict

import numpy as np

_SPECS: Dict[str, dict] = {}


def register(id: str, entry_point: Callable, max_episode_steps: int, kwargs: dict):
    _SPECS[id] = dict(id=id, entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=kwargs)


def register_env_variant(env_id: str, variants: dict, variant_id: str):
    """Deep-merge `variants` into a registered env's kwargs (env_variants.py:91-129)."""
    assert env_id in _SPECS, f"ERROR: {env_id} not found in env registry"
    spec = copy.deepcopy(_SPECS[env_id])
    variants = dict(variants)
    if "max_episode_steps" in variants:
        spec["max_episode_steps"] = variants.pop("max_episode_steps")

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v
    merge(spec["kwargs"], variants)
    spec["id"] = variant_id
    _SPECS[variant_id] = spec
    return variant_id


def register_env_with_variants(id, entry_point, max_episode_steps, kwargs):
    register(id, entry_point, max_episode_steps, kwargs)
    if id[:3] == "myo":
        register_env_variant(id, {"muscle_condition": "sarcopenia"}, id[:3] + "Sarc" + id[3:])
        register_env_variant(id, {"muscle_condition": "fatigue"}, id[:3] + "Fati" + id[3:])
    if id[:7] == "myoHand":
        register_env_variant(id, {"muscle_condition": "reafferentation"}, id[:3] + "Reaf" + id[3:])


def registry_specs():
    return _SPECS


def spec(id: str) -> dict:
    return _SPECS[id]


def make(id: str, num_envs: int = 1, device=None, seed=None, **overrides):
    """gym.make() equivalent returning a batched env (see envs/pose_v0.py)."""
    if id not in _SPECS:
        raise KeyError(f"unknown env id {id!r}; known: {sorted(_SPECS)[:8]} ...")
    s = _SPECS[id]
    kw = copy.deepcopy(s["kwargs"])
    kw.update(overrides)
    horizon = kw.pop("max_episode_steps", s["max_episode_steps"])      # gym.make(id, max_episode_steps=...) override
    env = s["entry_point"](env_id=id, num_envs=num_envs, device=device, seed=seed, max_episode_steps=horizon, **kw)
    env._make_args = dict(id=id, num_envs=num_envs, device=None if device is None else str(device), seed=seed, overrides=copy.deepcopy(overrides))
    return env


def _remake(args: dict):
    """unpickling: the env is rebuilt by its constructor (the reference's gym.utils.EzPickle), not copied buffer by buffer"""
    return make(args["id"], num_envs=args["num_envs"], device=args["device"], seed=args["seed"], **args["overrides"])


# ------------------------------------------------------------------ registrations
def _pose(**kw):
    from .pose_v0 import PoseEnvV0
    return PoseEnvV0(**kw)


def _torso(**kw):
    from .torso_v0 import TorsoEnvV0
    return TorsoEnvV0(**kw)


def _reach(**kw):
    from .reach_v0 import ReachEnvV0
    return ReachEnvV0(**kw)


def _reorient(**kw):
    from .reorient_v0 import ReorientEnvV0
    return ReorientEnvV0(**kw)


def _pen(**kw):
    from .reorient_v0 import PenTwirlEnvV0
    return PenTwirlEnvV0(**kw)


def _objhold(**kw):
    from .obj_hold_v0 import ObjHoldEnvV0
    return ObjHoldEnvV0(**kw)


def _stand(**kw):
    from .stand_v0 import StandEnvV0
    return StandEnvV0(**kw)


def _keyturn(**kw):
    from .key_turn_v0 import KeyTurnEnvV0
    return KeyTurnEnvV0(**kw)


def _walk(**kw):
    from .walk_v0 import WalkEnvV0
    return WalkEnvV0(**kw)


# Finger-tip reaching (myobase/__init__.py:55-105); motorFinger = the torque-motor counterpart (no "myo" prefix, so no
# muscle-condition variants: env_variants / __init__.py:13-49)
for _pre, _mdl, _h, _kw in (("motor", "motorfinger", 200, {"frame_skip": 5}), ("myo", "finger", 100, {})):
    register_env_with_variants(
        id=_pre + "FingerReachFixed-v0", entry_point=_reach, max_episode_steps=_h,
        kwargs=dict({"model": _mdl, "target_reach_range": {"IFtip": ((0.2, 0.05, 0.20), (0.2, 0.05, 0.20))},
                     "normalize_act": True, "target_frame": "world"}, **_kw))
    register_env_with_variants(
        id=_pre + "FingerReachRandom-v0", entry_point=_reach, max_episode_steps=_h,
        kwargs=dict({"model": _mdl, "target_reach_range": {"IFtip": ((0.1, -0.1, 0.1), (0.27, 0.1, 0.3))},
                     "normalize_act": True, "target_frame": "world"}, **_kw))
    # Finger-joint posing (myobase/__init__.py:188-254)
    register_env_with_variants(
        id=_pre + "FingerPoseFixed-v0", entry_point=_pose, max_episode_steps=_h,
        kwargs=dict({"model": _mdl, "target_jnt_range": {"IFadb": (0, 0), "IFmcp": (0, 0), "IFpip": (0.75, 0.75),
                                                         "IFdip": (0.75, 0.75)},
                     "viz_site_targets": ("IFtip",), "normalize_act": True}, **_kw))
    register_env_with_variants(
        id=_pre + "FingerPoseRandom-v0", entry_point=_pose, max_episode_steps=_h,
        kwargs=dict({"model": _mdl, "target_jnt_range": {"IFadb": (-0.2, 0.2), "IFmcp": (-0.4, 1), "IFpip": (0.1, 1),
                                                         "IFdip": (0.1, 1)},
                     "viz_site_targets": ("IFtip",), "normalize_act": True}, **_kw))

# Elbow posing (myobase/__init__.py:108-138)
register_env_with_variants(
    id="myoElbowPose1D6MFixed-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "elbow", "target_jnt_range": {"r_elbow_flex": (2, 2)}, "viz_site_targets": ("wrist",),
            "normalize_act": True, "pose_thd": 0.175, "reset_type": "random"})
register_env_with_variants(
    id="myoElbowPose1D6MRandom-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "elbow", "target_jnt_range": {"r_elbow_flex": (0, 2.27)}, "viz_site_targets": ("wrist",),
            "normalize_act": True, "pose_thd": 0.175, "reset_type": "random"})

# Elbow + exoskeleton posing (myobase/__init__.py:141-186): one extra torque actuator; the Random variant re-draws the
# mass of the carried weight per episode (pose_v0.py:177-187)
_EXO_W = {"pose": 1.0, "bonus": 4.0, "act_reg": 5.0, "penalty": 50}
register_env_with_variants(
    id="myoElbowPose1D6MExoFixed-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "elbow_exo", "target_jnt_range": {"r_elbow_flex": (2, 2)}, "viz_site_targets": ("wrist",),
            "normalize_act": True, "pose_thd": 0.175, "reset_type": "random", "weighted_reward_keys": _EXO_W})
register_env_with_variants(
    id="myoElbowPose1D6MExoRandom-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "elbow_exo", "target_jnt_range": {"r_elbow_flex": (0, 2.27)}, "viz_site_targets": ("wrist",),
            "normalize_act": True, "pose_thd": 0.175, "reset_type": "random", "weight_bodyname": "carry_weight",
            "weight_range": (0.1, 2), "weighted_reward_keys": _EXO_W})

# Hand ASL posing (myobase/__init__.py:300-415)
jnt_namesHand = ["pro_sup", "deviation", "flexion", "cmc_abduction", "cmc_flexion", "mp_flexion", "ip_flexion",
                 "mcp2_flexion", "mcp2_abduction", "pm2_flexion", "md2_flexion", "mcp3_flexion", "mcp3_abduction",
                 "pm3_flexion", "md3_flexion", "mcp4_flexion", "mcp4_abduction", "pm4_flexion", "md4_flexion",
                 "mcp5_flexion", "mcp5_abduction", "pm5_flexion", "md5_flexion"]
ASL_qpos = {
    0: "0 0 0 0.5624 0.28272 -0.75573 -1.309 1.30045 -0.006982 1.45492 0.998897 1.26466 0 1.40604 0.227795 1.07614 -0.020944 1.46103 0.06284 0.83263 -0.14399 1.571 1.38248",
    1: "0 0 0 0.0248 0.04536 -0.7854 -1.309 0.366605 0.010473 0.269258 0.111722 1.48459 0 1.45318 1.44532 1.44532 -0.204204 1.46103 1.44532 1.48459 -0.2618 1.47674 1.48459",
    2: "0 0 0 0.0248 0.04536 -0.7854 -1.13447 0.514973 0.010473 0.128305 0.111722 0.510575 0 0.37704 0.117825 1.44532 -0.204204 1.46103 1.44532 1.48459 -0.2618 1.47674 1.48459",
    3: "0 0 0 0.3384 0.25305 0.01569 -0.0262045 0.645885 0.010473 0.128305 0.111722 0.510575 0 0.37704 0.117825 1.571 -0.036652 1.52387 1.45318 1.40604 -0.068068 1.39033 1.571",
    4: "0 0 0 0.6392 -0.147495 -0.7854 -1.309 0.637158 0.010473 0.128305 0.111722 0.510575 0 0.37704 0.117825 0.306345 -0.010472 0.400605 0.133535 0.21994 -0.068068 0.274925 0.01571",
    5: "0 0 0 0.3384 0.25305 0.01569 -0.0262045 0.645885 0.010473 0.128305 0.111722 0.510575 0 0.37704 0.117825 0.306345 -0.010472 0.400605 0.133535 0.21994 -0.068068 0.274925 0.01571",
    6: "0 0 0 0.6392 -0.147495 -0.7854 -1.309 0.637158 0.010473 0.128305 0.111722 0.510575 0 0.37704 0.117825 0.306345 -0.010472 0.400605 0.133535 1.1861 -0.2618 1.35891 1.48459",
    7: "0 0 0 0.524 0.01569 -0.7854 -1.309 0.645885 -0.006982 0.128305 0.111722 0.510575 0 0.37704 0.117825 1.28036 -0.115192 1.52387 1.45318 0.432025 -0.068068 0.18852 0.149245",
    8: "0 0 0 0.428 0.22338 -0.7854 -1.309 0.645885 -0.006982 0.128305 0.194636 1.39033 0 1.08399 0.573415 0.667675 -0.020944 0 0.06284 0.432025 -0.068068 0.18852 0.149245",
    9: "0 0 0 0.5624 0.28272 -0.75573 -1.309 1.30045 -0.006982 1.45492 0.998897 0.39275 0 0.18852 0.227795 0.667675 -0.020944 0 0.06284 0.432025 -0.068068 0.18852 0.149245",
}
ASL_qpos = {k: np.array(v.split(" "), "float") for k, v in ASL_qpos.items()}
_HAND_SITES = ("THtip", "IFtip", "MFtip", "RFtip", "LFtip")
for k in ASL_qpos:
    register_env_with_variants(
        id="myoHandPose" + str(k) + "Fixed-v0", entry_point=_pose, max_episode_steps=100,
        kwargs={"model": "hand", "viz_site_targets": _HAND_SITES, "target_jnt_value": ASL_qpos[k],
                "normalize_act": True, "pose_thd": 0.7, "reset_type": "init", "target_type": "fixed"})
# myobase/__init__.py:259-297 ("Remove this when the ASL envs stablizes")
register_env_with_variants(
    id="myoHandPoseFixed-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "hand", "viz_site_targets": _HAND_SITES,
            "target_jnt_value": np.array([0, 0, 0, -0.0904, 0.0824475, -0.681555, -0.514888, 0, -0.013964, -0.0458132, 0,
                                          0.67553, -0.020944, 0.76979, 0.65982, 0, 0, 0, 0, 0.479155, -0.099484, 0.95831, 0]),
            "normalize_act": True, "pose_thd": 0.7, "reset_type": "init", "target_type": "fixed"})
_m = np.array([ASL_qpos[i] for i in range(10)]).astype(float)
Rpos = {n: (float(np.min(_m[:, i])), float(np.max(_m[:, i]))) for i, n in enumerate(jnt_namesHand)}
register_env_with_variants(
    id="myoHandPoseRandom-v0", entry_point=_pose, max_episode_steps=100,
    kwargs={"model": "hand", "viz_site_targets": _HAND_SITES, "target_jnt_range": Rpos, "normalize_act": True,
            "pose_thd": 0.7, "reset_type": "random", "target_type": "generate"})


# Hand-tip reaching (myobase/__init__.py:521-575).  `target_center` = the Fixed env's targets: the reference box centres,
# used by the synthetic-model env to re-centre the boxes on the synthetic hand's tips (reach_v0.py docstring).
_REACH_CENTER = {"THtip": (-0.165, -0.537, 1.495), "IFtip": (-0.151, -0.547, 1.455), "MFtip": (-0.146, -0.547, 1.447),
                 "RFtip": (-0.148, -0.543, 1.445), "LFtip": (-0.148, -0.528, 1.434)}
register_env_with_variants(
    id="myoHandReachFixed-v0", entry_point=_reach, max_episode_steps=100,
    kwargs={"model": "hand", "target_reach_range": {k: (v, v) for k, v in _REACH_CENTER.items()},
            "target_center": _REACH_CENTER, "normalize_act": True, "far_th": 0.044})
register_env_with_variants(
    id="myoHandReachRandom-v0", entry_point=_reach, max_episode_steps=100,
    kwargs={"model": "hand", "target_center": _REACH_CENTER, "normalize_act": True, "far_th": 0.034,
            "target_reach_range": {
                "THtip": ((-0.165 - 0.020, -0.537 - 0.040, 1.495 - 0.040), (-0.165 + 0.040, -0.537 + 0.020, 1.495 + 0.040)),
                "IFtip": ((-0.151 - 0.040, -0.547 - 0.020, 1.455 - 0.010), (-0.151 + 0.040, -0.547 + 0.020, 1.455 + 0.010)),
                "MFtip": ((-0.146 - 0.040, -0.547 - 0.020, 1.447 - 0.010), (-0.146 + 0.040, -0.547 + 0.020, 1.447 + 0.010)),
                "RFtip": ((-0.148 - 0.040, -0.543 - 0.020, 1.445 - 0.010), (-0.148 + 0.040, -0.543 + 0.020, 1.445 + 0.010)),
                "LFtip": ((-0.148 - 0.040, -0.528 - 0.020, 1.434 - 0.010), (-0.148 + 0.040, -0.528 + 0.020, 1.434 + 0.010))}})


# Leg standing (myobase/__init__.py:422-440): the reach task of walk_v0.py on the leg model
register_env_with_variants(
    id="myoLegStandRandom-v0", entry_point=_stand, max_episode_steps=150,
    kwargs={"model": "leg", "joint_random_range": (-0.2, 0.2), "target_reach_range": {"pelvis": ((-0.05, -0.05, 0), (0.05, 0.05, 0))},
            "normalize_act": True, "far_th": 0.44})

# Gait: torso walking (myobase/__init__.py:442-458).  The terrain variants (Rough/Hilly/Stair, :460-520) need
# height-field collision and are not registered.
register_env_with_variants(
    id="myoLegWalk-v0", entry_point=_walk, max_episode_steps=1000,
    kwargs={"model": "leg", "normalize_act": True, "min_height": 0.8, "max_rot": 0.8, "hip_period": 100,
            "reset_type": "init", "target_x_vel": 0.0, "target_y_vel": 1.2, "target_rot": None})


# Hand key turn (myobase/__init__.py:572-592)
register_env_with_variants(
    id="myoHandKeyTurnFixed-v0", entry_point=_keyturn, max_episode_steps=200,
    kwargs={"model": "hand_keyturn", "normalize_act": True})
register_env_with_variants(
    id="myoHandKeyTurnRandom-v0", entry_point=_keyturn, max_episode_steps=200,
    kwargs={"model": "hand_keyturn", "normalize_act": True, "key_init_range": (-np.pi / 2, np.pi / 2), "goal_th": 2 * np.pi})

# MyoTorso posing (myobase/__init__.py:638-701); myoTorsoExoPoseFixed-v0 runs on the torso with a synthetic back exosuit
# (two elastic cables with tension actuators: synth.make_torso(exosuit=True); the reference's myotorso_exosuit.xml is absent).
from ..model.synth import TORSO_JOINTS as _TJ
register_env_with_variants(
    id="myoTorsoPoseFixed-v0", entry_point=_torso, max_episode_steps=200,
    kwargs={"model": "torso", "normalize_act": True, "frame_skip": 5,
            "target_jnt_range": {j: ((-0.1, 0.1) if j == "lat_bending" else (0.0, 0.0)) for j in _TJ}})

register_env_with_variants(
    id="myoTorsoExoPoseFixed-v0", entry_point=_torso, max_episode_steps=200,
    kwargs={"model": "torso_exo", "normalize_act": True, "frame_skip": 5,
            "target_jnt_range": {j: ((-0.1, 0.1) if j == "lat_bending" else (0.0, 0.0)) for j in _TJ}})

# SAR reorient (myobase/__init__.py:703-749): frame_skip 5, horizon 50.
register_env_with_variants(
    id="myoHandReorient8-v0", entry_point=_reorient, max_episode_steps=50,
    kwargs={"model": "hand_reorient", "normalize_act": True, "frame_skip": 5, "geometries": "8"})
register_env_with_variants(
    id="myoHandReorient100-v0", entry_point=_reorient, max_episode_steps=50,
    kwargs={"model": "hand_reorient", "normalize_act": True, "frame_skip": 5, "geometries": "100"})
register_env_with_variants(
    id="myoHandReorientID-v0", entry_point=_reorient, max_episode_steps=50,
    kwargs={"model": "hand_reorient", "normalize_act": True, "frame_skip": 5, "geometries": "ID"})
register_env_with_variants(
    id="myoHandReorientOOD-v0", entry_point=_reorient, max_episode_steps=50,
    kwargs={"model": "hand_reorient", "normalize_act": True, "frame_skip": 5, "geometries": "OOD"})

# Pen twirl (myobase/__init__.py:615-634): frame_skip 5, horizon 50
register_env_with_variants(
    id="myoHandPenTwirlFixed-v0", entry_point=_pen, max_episode_steps=50,
    kwargs={"model": "hand_pen", "normalize_act": True, "frame_skip": 5, "random_target": False})
register_env_with_variants(
    id="myoHandPenTwirlRandom-v0", entry_point=_pen, max_episode_steps=50,
    kwargs={"model": "hand_pen", "normalize_act": True, "frame_skip": 5, "random_target": True})

# Hold objects (myobase/__init__.py:595-612): horizon 75
register_env_with_variants(
    id="myoHandObjHoldFixed-v0", entry_point=_objhold, max_episode_steps=75,
    kwargs={"model": "hand_hold", "normalize_act": True, "randomize": False})
register_env_with_variants(
    id="myoHandObjHoldRandom-v0", entry_point=_objhold, max_episode_steps=75,
    kwargs={"model": "hand_hold", "normalize_act": True, "randomize": True})
