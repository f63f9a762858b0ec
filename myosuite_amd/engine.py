"""ctypes binding of libmyosim_hip.so (C ABI in include/myosim.h).

PyTorch-ROCm is used only as plumbing: it owns the device buffers (tensors), the
current HIP stream and (in dist.py) the process group.  All per-step arithmetic
runs in the hand-written HIP kernels of myosuite_amd/csrc/myosim_engine.hip.
There is NO CPU fallback: if the library is missing or a call fails the error is
raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("MYOSIM_LIB", os.path.join(CSRC, "libmyosim_hip.so"))   # override only for A/B experiments
# sources: every *.hip under csrc/ (see build())
_lib = None

MM_TASK_NONE, MM_TASK_POSE, MM_TASK_REACH, MM_TASK_REORIENT, MM_TASK_WALK, MM_TASK_OBJHOLD, MM_TASK_KEYTURN = 0, 1, 2, 3, 4, 5, 6
RWD_KEYS_POSE = ["pose", "bonus", "penalty", "act_reg", "sparse", "solved", "done", "dense"]
RWD_KEYS_REACH = ["reach", "bonus", "penalty", "act_reg", "sparse", "solved", "done", "dense"]
RWD_KEYS_OBJHOLD = ["goal_dist", "bonus", "penalty", "act_reg", "sparse", "solved", "done", "dense"]
RWD_KEYS_KEYTURN = ["key_turn", "IFtip_approach", "THtip_approach", "act_reg", "bonus", "penalty", "sparse", "solved", "done",
                    "dense"]
RWD_KEYS_REORIENT = ["pos_align", "rot_align", "act_reg", "drop", "bonus", "sparse", "solved", "done", "dense"]
RWD_KEYS_WALK = ["vel_reward", "cyclic_hip", "ref_rot", "joint_angle_rew", "act_mag", "sparse", "solved", "done", "dense"]
(INFO_NQ, INFO_NV, INFO_NU, INFO_NA, INFO_NBODY, INFO_NSITE, INFO_NTENDON, INFO_LANES, INFO_LDS_PER_ENV,
 INFO_ENVS_PER_BLOCK, INFO_NGEOM, INFO_WAVES_PER_BLOCK, INFO_KERNEL_FAMILY, INFO_MODEL_WORDS,
 INFO_BODY_CHAINS, INFO_FOLDED_RESET, INFO_FWD_CARRY, INFO_TENDON_ITEMS, INFO_TENDON_FOLDED) = range(19)


MM_ABI_VERSION = 7   # include/myosim.h
MM_PREC_F32, MM_PREC_F64, MM_PREC_F64_STATE = 0, 1, 2   # mm_model_set_option("precision", ...)


class EngineError(RuntimeError):
    pass


# fp32 divide / sqrt as v_rcp / v_sqrt + one refinement (<= 2.5 ulp) instead of the correctly rounded expansions, hardware
# exp / log / sin / cos, reassociation and contraction: +4 % (hand) .. +5 % (leg-walk) env-steps/s with every parity test
# unchanged.  Infinities and NaNs stay honoured (far_th = inf, the bad-state check).  Group reductions do not depend on the
# flags: gsum() hides its operand from the optimiser and its stages are separate instructions.
EXTRA_FLAGS = os.environ.get("MYOSIM_HIPCC_FLAGS",
                             "-fno-hip-fp32-correctly-rounded-divide-sqrt -ffast-math -fhonor-infinities -fhonor-nans "
                             "-fno-slp-vectorize").split()
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations into v_pk_fma_f32 / v_pk_add_f32, whose operands must sit
# in aligned register pairs; in these kernels (every lane holds a few hundred live scalars) that costs more v_mov shuffles and
# spills than the packed instructions save: hand 5.05 -> 5.53 M env-steps/s, the other kernels +1..3 %.


# Machine-scheduler strategy per kernel group (measured on MI355X, env-steps/s, default scheduler -> chosen):
#   hand pose   <32,24>      4.07 M -> 4.15 M  iterative-maxocc
#   reorient    <64,32,GEN>  1.79 M -> 1.85 M  iterative-maxocc
#   leg walk    <64,40,GEN>  0.71 M -> 0.79 M  iterative-ilp      (iterative-maxocc: 0.71 M; iterative-minreg loses 10-25 % everywhere)
#   implicitfast leg <64,36,GEN,2>  kernel 0.753 -> 0.724 ms  iterative-ilp   (round 3; the self-contact hand <64,24,GEN> loses 2 % with it)
SCHED_STRATEGY = {"default": "iterative-maxocc", "myosim_inst_E.hip": "iterative-ilp"}
# (inst_H, the Euler leg: iterative-ilp until the stage fences went in; since then maxocc +2 %.  inst_J, the implicitfast units: iterative-ilp
#  until round 6 -- with the start rule of MM_SKIP_QACCSM in, ilp spills 2 / 8 VGPRs (12 / 36 B of scratch per lane) in the 36-wide leg
#  kernel where maxocc spills none at the same speed, 2.06 vs 2.06 M env-steps/s in one session; its SGPR spills, to VGPR lanes, go 217 -> 354)
# Extra per-file flags.  -sink-insts-to-avoid-spills: hand pose <32,24> 5.55 -> 5.67 M; within +-1 % (mostly -) on the others.
FILE_FLAGS = {"myosim_inst_B.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1", "-mllvm", "-amdgpu-set-wave-priority=1"],   # wave priority: hand +0.6 %
              # leg <64,36,GEN>: +0.8 %.  (Incremental Newton, -DMM_NEWTON_INCR=2 -- rank-one factor modifications when <= 2 rows changed set --
              # measured +2.4 % in the upright phase / +0.9 % in the steady mix of the episode, kernel 568 -> 559 us, for 52 spilled VGPRs and
              # 3x the HBM write traffic (4.0 -> 12.2 MB of scratch per launch): off.)
              "myosim_inst_H.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1"],
              "myosim_inst_I.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1"],   # self-contact hand <64,24,GEN>: VGPR spills 14 -> 0 (end of round 3)
              # reorient <64,32,GEN>: -sink-insts-to-avoid-spills took its VGPR spills from 74 to 18 in round 3 (kernel 0.705 -> 0.686 ms);
              # the 10...17 that the forward carry brought back in round 4 were three live 64-bit carry-row pointers, gone in round 5
              # (Engine::carry_row re-derives the address): 0 spilled VGPRs, 0 B of scratch, HBM traffic 10.5 -> 6.4 MB per launch.
              # MM_REFOLD_CARRY: the reset-observation pass of a folded reset also writes a forward-carry row: reorient 3.95 -> 4.11 M in
              # one session (the leg unit, inst_H, loses 1 % with it: off there)
              "myosim_inst_D.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1", "-DMM_REFOLD_CARRY=1"],
              # implicitfast units: with the forward carry in (a second inlined implicit solve in the trailing pass) the 36-wide leg kernel
              # spilled 52 / 60 VGPRs (148 / 188 B of scratch per lane, 13.5 MB of scratch writes per launch); with the flag 0 / 0 and
              # +0.9 % (1.690 -> 1.705 M in one session).  Round 3 had measured -1 % for it on this unit, at 26 spills and no carry.
              # (incremental Newton measured +2.3 % here -- and 45 spilled VGPRs: off, this unit stays at zero)
              "myosim_inst_J.hip": ["-mllvm", "-sink-insts-to-avoid-spills=1"],
              # precision-mode (fp64) kernels: IEEE divide / sqrt and no reassociation -- these exist to track the fp64 reference;
              # fma contraction stays on (it only removes roundings)
              "myosim_inst_P.hip": ["-fno-fast-math", "-ffp-contract=fast"],
              "myosim_inst_Q.hip": ["-fno-fast-math", "-ffp-contract=fast"],
              "myosim_inst_R.hip": ["-fno-fast-math", "-ffp-contract=fast"]}
# (myosim_ppo.hip, the PPO learner kernels, takes the default flags: with IEEE exp / log / divide its loss stage alone was 16 k of a
#  workgroup's 131 k cycles; the gradient test against torch autograd holds its 2e-4 either way)


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """Compile the HIP engine for gfx950 in-tree (hipcc cross-compiles without a GPU).  The kernel instantiations are
    spread over several translation units (myosim_inst_*.hip) that are compiled in parallel and linked into one .so."""
    import concurrent.futures
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))] + \
           [os.path.join(_HERE, "..", "include", h) for h in ("myosim.h", "myosim_model.h")]
    ppo_hdr = os.path.join(_HERE, "..", "include", "myosim_ppo.h")         # seen by myosim_ppo.hip only
    deps = srcs + hdrs + [ppo_hdr]
    bdir = os.path.join(CSRC, "_build")
    # the compiler flags are part of the build's identity: a library built with other flags is rebuilt from scratch
    stamp, flags_now = os.path.join(bdir, "flags.txt"), " ".join(EXTRA_FLAGS) + " | " + repr(sorted(SCHED_STRATEGY.items())) + repr(sorted(FILE_FLAGS.items()))
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags_now
    force = force or (os.path.exists(LIB_PATH) and os.path.isdir(bdir) and not same_flags)
    if not force and os.path.exists(LIB_PATH) and (same_flags or not os.path.isdir(bdir)) and \
            all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    os.makedirs(bdir, exist_ok=True)
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)

    def compile_one(src):
        obj = os.path.join(bdir, os.path.basename(src)[:-4] + ".o")
        own_hdr = os.path.getmtime(ppo_hdr) if os.path.basename(src) == "myosim_ppo.hip" else 0.0
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_hdr, own_hdr):
            return obj
        sched = SCHED_STRATEGY.get(os.path.basename(src), SCHED_STRATEGY["default"])
        cmd = (["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS + FILE_FLAGS.get(os.path.basename(src), []) +
               (["-mllvm", f"-amdgpu-sched-strategy={sched}"] if sched and src.endswith(".hip") and "inst" in os.path.basename(src) else []) +
               ["-c", "-o", obj, src])
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(srcs), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(flags_now)
    return LIB_PATH


MM_PPO_MAX_LAYERS, MM_PPO_MAX_WIDTH, MM_PPO_MAX_OBS, MM_PPO_MAX_OUT = 8, 128, 512, 256        # include/myosim_ppo.h
MM_PPO_SQUASH_TANH, MM_PPO_SQUASH_SIGMOID = 0, 1


class mm_ppo_config(C.Structure):
    _fields_ = [("size", C.c_int), ("obs_dim", C.c_int), ("act_dim", C.c_int), ("pi_layers", C.c_int), ("vf_layers", C.c_int),
                ("pi_widths", C.c_int * MM_PPO_MAX_LAYERS), ("vf_widths", C.c_int * MM_PPO_MAX_LAYERS), ("squash", C.c_int),
                ("max_minibatch", C.c_int), ("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("adam_eps", C.c_float), ("clipping_epsilon", C.c_float), ("entropy_cost", C.c_float), ("value_cost", C.c_float),
                ("max_grad_norm", C.c_float)]


class mm_state(C.Structure):
    _fields_ = [("nenv", C.c_int), ("qpos", C.c_void_p), ("qvel", C.c_void_p), ("act", C.c_void_p),
                ("qacc_warmstart", C.c_void_p), ("time", C.c_void_p), ("status", C.c_void_p),
                ("geom_size_env", C.c_void_p), ("geom_env_id", C.c_int), ("geom_type_env", C.c_void_p),
                ("body_mass_env", C.c_void_p), ("body_mass_env_id", C.c_int),
                ("body_pos_env", C.c_void_p), ("body_pos_env_id", C.c_int), ("env_index_base", C.c_int)]


_DERIVED_FIELDS = ["xpos", "xquat", "xipos", "site_xpos", "geom_xpos", "cvel", "subtree_com", "actuator_length",
                   "actuator_velocity", "actuator_force", "qacc", "ten_length", "nefc", "solver_niter"]


class mm_derived(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in _DERIVED_FIELDS]


class mm_task(C.Structure):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.size = C.sizeof(mm_task)      # the library copies min(size, its sizeof) bytes: see include/myosim.h

    _fields_ = [("size", C.c_uint32), ("task", C.c_int), ("nsubsteps", C.c_int), ("normalize_act", C.c_int), ("do_forward", C.c_int),
                ("fatigue", C.c_int), ("max_episode_steps", C.c_int),
                ("pose_thd", C.c_float), ("far_th", C.c_float),
                ("w_pose", C.c_float), ("w_bonus", C.c_float), ("w_act_reg", C.c_float), ("w_penalty", C.c_float),
                ("target_jnt_value", C.c_void_p),
                ("fat_MA", C.c_void_p), ("fat_MR", C.c_void_p), ("fat_MF", C.c_void_p),
                ("fat_F", C.c_float), ("fat_R", C.c_float), ("fat_r", C.c_float),
                ("obs", C.c_void_p), ("obs_dim", C.c_int), ("rwd", C.c_void_p), ("done", C.c_void_p),
                ("truncated", C.c_void_p), ("step_count", C.c_void_p), ("ctrl_out", C.c_void_p),
                ("reaf_src", C.c_int), ("reaf_dst", C.c_int), ("obs_layout", C.c_int), ("act_reg_mean", C.c_int), ("obs_dt", C.c_float),
                ("tip_sites", C.c_void_p), ("ntip", C.c_int), ("target_pos", C.c_void_p), ("reach_far_th", C.c_float), ("reach_stand", C.c_int),
                ("walk_body", C.c_int * 4), ("walk_qadr", C.c_int * 6), ("walk_min_height", C.c_float),
                ("walk_max_rot", C.c_float), ("walk_hip_period", C.c_int), ("walk_target_x_vel", C.c_float),
                ("walk_target_y_vel", C.c_float), ("walk_target_rot", C.c_float * 4), ("walk_w", C.c_float * 5),
                ("reor_obj_body", C.c_int), ("reor_eps_site", C.c_int), ("reor_pen_length", C.c_float),
                ("reor_axis_half", C.c_void_p), ("reor_des_rot", C.c_void_p), ("reor_w", C.c_float * 5),
                ("reor_obs_muscle", C.c_int),
                ("key_goal_th", C.c_float), ("key_w", C.c_float * 6),
                ("env_mask", C.c_void_p), ("obs_only", C.c_int), ("fwd_carry", C.c_void_p)]


class mm_rollout(C.Structure):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.size = C.sizeof(mm_rollout)

    _fields_ = [("size", C.c_uint32), ("action", C.c_void_p), ("action_seed", C.c_uint64), ("action_stream", C.c_uint64), ("action_out", C.c_void_p),
                ("ep_stats", C.c_void_p), ("reset_mask", C.c_void_p), ("autoreset", C.c_int), ("random_qpos", C.c_int),
                ("qlo", C.c_void_p), ("qhi", C.c_void_p), ("tlo", C.c_void_p), ("thi", C.c_void_p), ("target", C.c_void_p),
                ("episode", C.c_void_p), ("reset_seed", C.c_uint64),
                ("walk_ka_qpos", C.c_void_p), ("walk_ka_qvel", C.c_void_p), ("walk_kb_qpos", C.c_void_p), ("walk_kb_qvel", C.c_void_p),
                ("walk_random", C.c_int),
                ("reor_init_qpos", C.c_void_p), ("reor_size_tables", C.c_void_p), ("reor_ntab", C.c_int), ("reor_tar_length", C.c_float),
                ("reor_geom_size_env", C.c_void_p), ("reor_geom_type_env", C.c_void_p), ("reor_axis_half", C.c_void_p),
                ("reor_des_rot", C.c_void_p), ("fat_reset_vec", C.c_void_p)]


def lib():
    """Load libmyosim_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                f"{LIB_PATH} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (needs hipcc). There is no CPU fallback for the physics step.")
        L = C.CDLL(LIB_PATH)
        L.mm_model_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.mm_model_destroy.argtypes = [C.c_void_p]
        L.mm_model_info.argtypes = [C.c_void_p, C.c_int]
        L.mm_model_set_lanes.argtypes = [C.c_void_p, C.c_int]
        L.mm_step.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_int, C.c_void_p]
        L.mm_forward.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.POINTER(mm_derived), C.c_void_p]
        L.mm_env_step.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.POINTER(mm_task),
                                  C.POINTER(mm_derived), C.c_void_p]
        L.mm_rollout_step.argtypes = [C.c_void_p, C.POINTER(mm_state), C.POINTER(mm_task), C.POINTER(mm_rollout),
                                      C.POINTER(mm_derived), C.c_void_p]
        L.mm_model_launch_lanes.argtypes = [C.c_void_p, C.c_int]
        L.mm_uniform_at.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        L.mm_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.mm_pose_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                    C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mm_uniform.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_void_p]
        L.mm_episode_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p]
        L.mm_gae.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.mm_fatigue_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.mm_env_draw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]
        L.mm_reach_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.mm_walk_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.mm_reorient_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.c_void_p]
        L.mm_reorient_reset_typed.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                              C.c_void_p, C.c_uint64, C.c_void_p]
        L.mm_pen_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float,
                                   C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.mm_objhold_reset.argtypes = [C.c_void_p, C.POINTER(mm_state), C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                       C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.mm_last_error.restype = C.c_char_p
        L.mm_version.restype = C.c_char_p
        L.mm_debug_layout.argtypes = [C.c_void_p, C.c_char_p]
        L.mm_debug_set_dump.argtypes = [C.c_void_p]
        L.mm_debug_set_prof.argtypes = [C.c_void_p]
        L.mm_model_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.mm_model_launch_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        # fused PPO learner kernels (include/myosim_ppo.h)
        L.mm_ppo_create.argtypes = [C.POINTER(mm_ppo_config), C.c_int, C.POINTER(C.c_void_p)]
        L.mm_ppo_destroy.argtypes = [C.c_void_p]
        L.mm_ppo_destroy.restype = None
        L.mm_ppo_last_error.restype = C.c_char_p
        L.mm_ppo_param_count.argtypes = [C.c_void_p]
        L.mm_ppo_value_offset.argtypes = [C.c_void_p]
        L.mm_ppo_reset_optimizer.argtypes = [C.c_void_p, C.c_void_p]
        L.mm_ppo_act.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 6
        L.mm_ppo_store.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
        L.mm_ppo_grad.argtypes = [C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 6
        L.mm_ppo_adam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        L.mm_ppo_set_entropy_noise.argtypes = [C.c_void_p, C.c_void_p]
        # the binding restates the header's structs: refuse a library built from another ABI or with other struct layouts
        L.mm_struct_size.argtypes = [C.c_int]
        if L.mm_abi_version() != MM_ABI_VERSION:
            raise EngineError(f"{LIB_PATH} speaks ABI {L.mm_abi_version()}, this binding ABI {MM_ABI_VERSION}: rebuild the library")
        for which, st in enumerate((mm_state, mm_derived, mm_task, mm_rollout)):
            if L.mm_struct_size(which) != C.sizeof(st):
                raise EngineError(f"struct layout mismatch: {st.__name__} is {L.mm_struct_size(which)} bytes in the library, "
                                  f"{C.sizeof(st)} in the binding (include/myosim.h changed without engine.py)")
        _lib = L
    return _lib


def _chk(rc: int, what: str):
    if rc != 0:
        raise EngineError(f"{what} failed (rc={rc}): {lib().mm_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "engine buffers must be contiguous device tensors"
    return t.data_ptr()


def _stream(device=None):
    """current torch stream of `device` (the model's / output tensor's device, not whatever device happens to be current)"""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class HipModel:
    """Device-resident compiled model (mm_model)."""

    def __init__(self, compiled, lanes_per_env: int = 0, device: Optional[torch.device] = None, precision: int = MM_PREC_F32):
        """precision: MM_PREC_F32 (default), MM_PREC_F64 (fp64 arithmetic, fp32 buffers) or MM_PREC_F64_STATE (fp64 state rows
        too: BatchState then allocates float64 qpos / qvel / act / qacc_warmstart); include/myosim.h"""
        self.cm = compiled
        self.precision = MM_PREC_F32
        self._n_states = 0          # BatchStates allocated for this model (their row width is fixed by the precision mode at that time)
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible: the physics step only runs on the GPU (no CPU fallback)")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        blob = np.ascontiguousarray(compiled.blob, dtype=np.uint32)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _chk(lib().mm_model_create(blob.ctypes.data, int(blob.size), C.byref(h)), "mm_model_create")
        self.h = h
        if lanes_per_env:       # (pinned first: the precision option then refuses a family that has no kernel at the pinned width)
            _chk(lib().mm_model_set_lanes(self.h, lanes_per_env), "mm_model_set_lanes")
        if precision != MM_PREC_F32:
            self.set_option("precision", precision)

    def info(self, which: int) -> int:
        return lib().mm_model_info(self.h, which)

    def launch_lanes(self, nenv: int) -> int:
        """lanes per env a launch over `nenv` envs uses (picked from the batch size unless pinned)"""
        return lib().mm_model_launch_lanes(self.h, int(nenv))

    LAUNCH_KEYS = ("lanes", "waves_per_block", "two_wave", "lds_model", "lds_bytes", "blocks", "resident_blocks_per_cu", "vgprs")

    def launch_info(self, nenv: int) -> dict:
        """geometry and occupancy of the env-step launch over `nenv` envs (mm_model_launch_info; nothing is launched)"""
        out = (C.c_int * len(self.LAUNCH_KEYS))()
        with torch.cuda.device(self.device):
            _chk(lib().mm_model_launch_info(self.h, int(nenv), out, len(self.LAUNCH_KEYS)), "mm_model_launch_info")
        return dict(zip(self.LAUNCH_KEYS, (int(v) for v in out)))

    def set_option(self, name: str, value: int):
        """mm_model_set_option.  "precision" also decides the element type of the state rows a BatchState allocates (fp64 under
        MM_PREC_F64_STATE): the mode is tracked here, and a change of the state-row width is refused once a BatchState of this
        model exists -- its buffers would be read and written at the wrong width."""
        if name == "precision":
            wide = lambda p: int(p) == MM_PREC_F64_STATE
            if self._n_states and wide(value) != wide(self.precision):
                raise EngineError("precision: the state-row width cannot change after a BatchState was created for this model "
                                  "(create the model with precision=... instead)")
        _chk(lib().mm_model_set_option(self.h, name.encode(), int(value)), "mm_model_set_option")
        if name == "precision":
            self.precision = int(value)

    def layout(self, name: str) -> int:
        return lib().mm_debug_layout(self.h, name.encode())

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib().mm_model_destroy(self.h)
        except Exception:
            pass


class BatchState:
    """Env-major state tensors of E environments (mm_state)."""

    def __init__(self, model: HipModel, nenv: int):
        cm = model.cm
        dev = model.device
        if int(nenv) < 1:
            raise ValueError(f"BatchState needs at least one environment (got nenv = {nenv}); the C ABI refuses an empty batch with MM_EARG")
        self.model = model
        self.nenv = nenv
        model._n_states += 1
        f = dict(dtype=torch.float32, device=dev)
        # the four state rows are float64 under MM_PREC_F64_STATE (qpos0 is an fp32 model table in every mode)
        fs = dict(dtype=torch.float64 if model.precision == MM_PREC_F64_STATE else torch.float32, device=dev)
        self.qpos = torch.from_numpy(np.tile(cm.qpos0.astype(np.float32), (nenv, 1))).to(dev).to(fs["dtype"]).contiguous()
        self.qvel = torch.zeros(nenv, cm.nv, **fs)
        self.act = torch.zeros(nenv, max(cm.na, 1), **fs)[:, :cm.na].contiguous() if cm.na == 0 else torch.zeros(nenv, cm.na, **fs)
        self.qacc_warmstart = torch.zeros(nenv, cm.nv, **fs)
        self.time = torch.zeros(nenv, **f)
        self.status = torch.zeros(nenv, dtype=torch.int32, device=dev)
        self.geom_size_env = None
        self._c = mm_state(nenv, _ptr(self.qpos), _ptr(self.qvel), self.act.data_ptr(), _ptr(self.qacc_warmstart),
                           _ptr(self.time), _ptr(self.status), None, -1, None, None, -1, None, -1, 0)
        self.geom_type_env = None
        self.body_mass_env = None
        self.body_pos_env = None

    def set_geom_size_env(self, geom_id: int, sizes: torch.Tensor):
        """per-env model delta: collision size [nenv][3] of one geom (mm_state.geom_size_env)"""
        assert sizes.shape == (self.nenv, 3) and sizes.dtype == torch.float32 and sizes.is_contiguous() and sizes.is_cuda
        self.geom_size_env = sizes
        self._c.geom_size_env = sizes.data_ptr(); self._c.geom_env_id = int(geom_id)

    def set_body_mass_env(self, body_id: int, mass: torch.Tensor):
        """per-env model delta: mass [nenv] of one body (mm_state.body_mass_env)"""
        assert mass.shape == (self.nenv,) and mass.dtype == torch.float32 and mass.is_contiguous()
        self.body_mass_env = mass
        self._c.body_mass_env = mass.data_ptr(); self._c.body_mass_env_id = int(body_id)

    def set_body_pos_env(self, body_id: int, pos: torch.Tensor):
        """per-env model delta: frame position [nenv][3] of one body in its parent (mm_state.body_pos_env)"""
        assert pos.shape == (self.nenv, 3) and pos.dtype == torch.float32 and pos.is_contiguous()
        self.body_pos_env = pos
        self._c.body_pos_env = pos.data_ptr(); self._c.body_pos_env_id = int(body_id)

    def set_geom_type_env(self, types: torch.Tensor):
        """per-env model delta: type [nenv] int32 of the geom named in set_geom_size_env (mm_state.geom_type_env)"""
        assert types.shape == (self.nenv,) and types.dtype == torch.int32 and types.is_contiguous() and types.is_cuda
        self.geom_type_env = types
        self._c.geom_type_env = types.data_ptr()

    @property
    def env_index_base(self) -> int:
        return int(self._c.env_index_base)

    @env_index_base.setter
    def env_index_base(self, v: int):
        """global index of env 0 of this shard: every Philox stream (resets, draws, in-kernel actions) is keyed by base + e"""
        self._c.env_index_base = int(v)

    @property
    def c(self):
        return C.byref(self._c)


class Derived:
    """Requested derived outputs of the final forward pass (mm_derived)."""

    def __init__(self, model: HipModel, nenv: int, fields):
        cm = model.cm
        dev = model.device
        shapes = {"xpos": (cm.nbody, 3), "xquat": (cm.nbody, 4), "xipos": (cm.nbody, 3), "site_xpos": (cm.nsite, 3),
                  "geom_xpos": (cm.ngeom, 3), "cvel": (cm.nbody, 6), "subtree_com": (cm.nbody, 3),
                  "actuator_length": (cm.nu,), "actuator_velocity": (cm.nu,), "actuator_force": (cm.nu,),
                  "qacc": (cm.nv,), "ten_length": (cm.ntendon,), "nefc": (), "solver_niter": ()}
        self.t: Dict[str, torch.Tensor] = {}
        self._c = mm_derived()
        for name in fields:
            dt = torch.int32 if name in ("nefc", "solver_niter") else torch.float32
            self.t[name] = torch.zeros((nenv,) + shapes[name], dtype=dt, device=dev)
            setattr(self._c, name, self.t[name].data_ptr())

    def __getitem__(self, k):
        return self.t[k]

    @property
    def c(self):
        return C.byref(self._c)


def step(model: HipModel, state: BatchState, ctrl: torch.Tensor, nsub: int = 1):
    """`nsub` raw mj_step substeps with ctrl [E,nu] applied unchanged."""
    assert ctrl.shape == (state.nenv, model.cm.nu) and ctrl.dtype == torch.float32
    _chk(lib().mm_step(model.h, state.c, _ptr(ctrl.contiguous()), int(nsub), _stream(model.device)), "mm_step")


def forward(model: HipModel, state: BatchState, ctrl: Optional[torch.Tensor] = None, derived: Optional[Derived] = None):
    _chk(lib().mm_forward(model.h, state.c, _ptr(ctrl) if ctrl is not None else None,
                          derived.c if derived is not None else None, _stream(model.device)), "mm_forward")


def env_step(model: HipModel, state: BatchState, action: Optional[torch.Tensor], task: mm_task,
             derived: Optional[Derived] = None):
    if action is not None:
        assert action.shape == (state.nenv, model.cm.nu) and action.dtype == torch.float32 and action.is_contiguous()
    else:
        assert task.obs_only, "action may only be omitted for an obs_only pass"
    _chk(lib().mm_env_step(model.h, state.c, _ptr(action), C.byref(task), derived.c if derived is not None else None,
                           _stream(model.device)), "mm_env_step")


def rollout_step(model: HipModel, state: BatchState, task: mm_task, ro: mm_rollout, derived: Optional[Derived] = None):
    """mm_env_step with the rollout bookkeeping of `ro` folded into the same launch (mm_rollout_step)."""
    _chk(lib().mm_rollout_step(model.h, state.c, C.byref(task), C.byref(ro), derived.c if derived is not None else None,
                               _stream(model.device)), "mm_rollout_step")


def reset_observation(model: HipModel, state: BatchState, task: mm_task, mask: Optional[torch.Tensor] = None):
    """Observation (and reward terms) of the CURRENT state of the masked envs: one forward pass, no stepping, no
    counters -- what MujocoEnv.reset returns after Robot.reset (env_base.py:560-575)."""
    t = mm_task.from_buffer_copy(task)
    t.obs_only = 1
    t.env_mask = _ptr(mask)
    env_step(model, state, None, t)


def reset(model: HipModel, state: BatchState, mask: Optional[torch.Tensor] = None, qpos: Optional[torch.Tensor] = None,
          qvel: Optional[torch.Tensor] = None):
    if mask is not None:
        assert mask.dtype == torch.uint8
    _chk(lib().mm_reset(model.h, state.c, _ptr(mask), _ptr(qpos), _ptr(qvel), _stream(model.device)), "mm_reset")


def pose_reset(model: HipModel, state: BatchState, mask, qlo, qhi, tlo, thi, target, episode, step_count, seed: int,
               random_qpos: bool, obs=None, obs_layout: int = 0):
    _chk(lib().mm_pose_reset(model.h, state.c, _ptr(mask), _ptr(qlo), _ptr(qhi), _ptr(tlo), _ptr(thi), _ptr(target),
                             _ptr(episode), _ptr(step_count), C.c_uint64(seed), int(random_qpos), _ptr(obs),
                             0 if obs is None else int(obs.shape[1]), int(obs_layout), _stream(model.device)),
         "mm_pose_reset")


def reach_reset(model: HipModel, state: BatchState, mask, tlo, thi, target, tip0, ntip: int, episode, step_count,
                seed: int, obs=None):
    _chk(lib().mm_reach_reset(model.h, state.c, _ptr(mask), _ptr(tlo), _ptr(thi), _ptr(target), _ptr(tip0), int(ntip),
                              _ptr(episode), _ptr(step_count), C.c_uint64(seed), _ptr(obs),
                              0 if obs is None else int(obs.shape[1]), _stream(model.device)), "mm_reach_reset")


def walk_reset(model: HipModel, state: BatchState, mask, key_a_qpos, key_a_qvel, key_b_qpos, key_b_qvel, random: bool,
               episode, step_count, seed: int):
    _chk(lib().mm_walk_reset(model.h, state.c, _ptr(mask), _ptr(key_a_qpos), _ptr(key_a_qvel), _ptr(key_b_qpos),
                             _ptr(key_b_qvel), int(random), _ptr(episode), _ptr(step_count), C.c_uint64(seed), _stream(model.device)),
         "mm_walk_reset")


def reorient_reset(model: HipModel, state: BatchState, mask, init_qpos, size_table, axis_half, des_rot, tar_length: float,
                   episode, step_count, seed: int):
    assert state.geom_size_env is not None, "call BatchState.set_geom_size_env first"
    _chk(lib().mm_reorient_reset(model.h, state.c, _ptr(mask), _ptr(init_qpos), _ptr(size_table), int(size_table.shape[0]),
                                 _ptr(state.geom_size_env), _ptr(axis_half), _ptr(des_rot), C.c_float(tar_length),
                                 _ptr(episode), _ptr(step_count), C.c_uint64(seed), _stream(model.device)), "mm_reorient_reset")


def reorient_reset_typed(model: HipModel, state: BatchState, mask, init_qpos, size_tables, axis_half, des_rot,
                         tar_length: float, episode, step_count, seed: int):
    """size_tables [4][ntab][3]: capsule, ellipsoid, cylinder, box"""
    assert state.geom_size_env is not None and state.geom_type_env is not None and size_tables.shape[0] == 4
    _chk(lib().mm_reorient_reset_typed(model.h, state.c, _ptr(mask), _ptr(init_qpos), _ptr(size_tables),
                                       int(size_tables.shape[1]), _ptr(state.geom_size_env), _ptr(state.geom_type_env),
                                       _ptr(axis_half), _ptr(des_rot), C.c_float(tar_length), _ptr(episode),
                                       _ptr(step_count), C.c_uint64(seed), _stream(model.device)), "mm_reorient_reset_typed")


def pen_reset(model: HipModel, state: BatchState, mask, init_qpos, axis_half: float, ranges, des_rot, tar_length: float,
              episode, step_count, seed: int):
    """ranges = (lo0, hi0, lo1, hi1) of desired_orien[0:2] (pen_v0.py:175-178); zeros for the Fixed task"""
    _chk(lib().mm_pen_reset(model.h, state.c, _ptr(mask), _ptr(init_qpos), C.c_float(axis_half), *[C.c_float(x) for x in ranges],
                            _ptr(des_rot), C.c_float(tar_length), _ptr(episode), _ptr(step_count), C.c_uint64(seed), _stream(model.device)),
         "mm_pen_reset")


def objhold_reset(model: HipModel, state: BatchState, mask, init_qpos, goal_center, goal_half: float, size_range, goal,
                  episode, step_count, seed: int):
    """size_range (lo, hi) re-draws the object size into state.geom_size_env (Random task); None keeps the model's size"""
    gs = state.geom_size_env if size_range is not None else None
    lo, hi = size_range if size_range is not None else (0.0, 0.0)
    _chk(lib().mm_objhold_reset(model.h, state.c, _ptr(mask), _ptr(init_qpos), _ptr(goal_center), C.c_float(goal_half),
                                C.c_float(lo), C.c_float(hi), _ptr(goal), _ptr(gs), _ptr(episode), _ptr(step_count),
                                C.c_uint64(seed), _stream(model.device)), "mm_objhold_reset")


def env_draw(out: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor, mask, episode, seed: int, stream_id: int, base=None,
             env_index_base: int = 0):
    """out[e, k] = base[k] + lo[k] + (hi[k]-lo[k]) * U[0,1): per-episode draw of a per-env model delta (mm_env_draw); call before
    the task reset of the same episode."""
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
    n, k = out.shape[0], (out.shape[1] if out.dim() > 1 else 1)
    assert lo.numel() == k and hi.numel() == k and (base is None or base.numel() == k)
    _chk(lib().mm_env_draw(out.data_ptr(), n, k, _ptr(base), _ptr(lo), _ptr(hi), _ptr(mask), _ptr(episode), C.c_uint64(seed),
                           C.c_uint32(stream_id), int(env_index_base), _stream(out.device)), "mm_env_draw")
    return out


def fatigue_reset(MA: torch.Tensor, MR: torch.Tensor, MF: torch.Tensor, mask, vec=None):
    """3CC-r state of the masked envs back to rest in one launch (fatigue.py:82-99): MF = vec or 0, MR = 1 - MF, MA = 0"""
    assert MA.shape == MR.shape == MF.shape and MA.is_contiguous() and MR.is_contiguous() and MF.is_contiguous()
    _chk(lib().mm_fatigue_reset(_ptr(MA), _ptr(MR), _ptr(MF), _ptr(mask), _ptr(vec), int(MA.shape[0]), int(MA.shape[1]),
                                _stream(MA.device)), "mm_fatigue_reset")


def episode_stats(stats: torch.Tensor, reset_mask: torch.Tensor, rwd: torch.Tensor, dense_col: int, solved_col: int,
                  done: torch.Tensor, truncated: torch.Tensor):
    """stats[e] = (return, length, solved) accumulated with this step's reward row; reset_mask = done | truncated (one launch)."""
    assert stats.shape == (rwd.shape[0], 3) and stats.dtype == torch.float32 and reset_mask.dtype == torch.uint8
    _chk(lib().mm_episode_stats(_ptr(stats), _ptr(reset_mask), _ptr(rwd), int(rwd.shape[1]), int(dense_col), int(solved_col),
                                _ptr(done), _ptr(truncated), int(rwd.shape[0]), _stream(stats.device)), "mm_episode_stats")


def gae(reward: torch.Tensor, terminated: torch.Tensor, truncated: Optional[torch.Tensor], value: torch.Tensor, advantage: torch.Tensor,
        returns: torch.Tensor, gamma: float, lam: float):
    """GAE of a [T, n] unroll in one launch (mm_gae): value is [T + 1, n]; advantage / returns are written."""
    T, n = reward.shape
    assert value.shape == (T + 1, n) and advantage.shape == (T, n) and returns.shape == (T, n)
    for t in (reward, terminated, value, advantage, returns) + ((truncated,) if truncated is not None else ()):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    _chk(lib().mm_gae(_ptr(reward), _ptr(terminated), _ptr(truncated), _ptr(value), _ptr(advantage), _ptr(returns), int(T), int(n),
                      C.c_float(gamma), C.c_float(lam), _stream(reward.device)), "mm_gae")


class FusedPPO:
    """Handle of the fused PPO learner kernels (include/myosim_ppo.h): policy / value MLPs over ONE flat parameter vector (policy
    first; per layer weight [out][in], bias [out] -- the parameter order of torch nn.Sequential(Linear, SiLU, ...)).

    act():   one launch per rollout step (normalise, policy forward, sample, log-prob, squash, value forward)
    grad():  two launches per minibatch (gather + forward + losses + backward into per-workgroup partials; ordered reduction)
    adam():  one launch (global-norm clip + Adam), two after a data-parallel all-reduce (norm recomputed)."""

    def __init__(self, obs_dim: int, act_dim: int, policy_hidden, value_hidden, squash: str, max_minibatch: int, learning_rate: float,
                 clipping_epsilon: float, entropy_cost: float, value_cost: float, max_grad_norm: Optional[float],
                 device: Optional[torch.device] = None, betas=(0.9, 0.999), adam_eps: float = 1e-8):
        if not torch.cuda.is_available():
            raise EngineError("no HIP device visible: the fused PPO kernels only run on the GPU")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        c = mm_ppo_config()
        c.size = C.sizeof(mm_ppo_config)
        c.obs_dim, c.act_dim = int(obs_dim), int(act_dim)
        pw, vw = list(policy_hidden) + [2 * act_dim], list(value_hidden) + [1]
        if len(pw) > MM_PPO_MAX_LAYERS or len(vw) > MM_PPO_MAX_LAYERS:
            raise EngineError(f"fused PPO kernels take at most {MM_PPO_MAX_LAYERS} linear layers per network")
        c.pi_layers, c.vf_layers = len(pw), len(vw)
        for i, w in enumerate(pw):
            c.pi_widths[i] = int(w)
        for i, w in enumerate(vw):
            c.vf_widths[i] = int(w)
        c.squash = {"tanh": MM_PPO_SQUASH_TANH, "sigmoid": MM_PPO_SQUASH_SIGMOID}[squash]
        c.max_minibatch = int(max_minibatch)
        c.learning_rate, c.beta1, c.beta2, c.adam_eps = float(learning_rate), float(betas[0]), float(betas[1]), float(adam_eps)
        c.clipping_epsilon, c.entropy_cost, c.value_cost = float(clipping_epsilon), float(entropy_cost), float(value_cost)
        c.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self.cfg = c
        h = C.c_void_p()
        self._chk(lib().mm_ppo_create(C.byref(c), self.device.index or 0, C.byref(h)), "mm_ppo_create")
        self.h = h
        self.param_count = lib().mm_ppo_param_count(h)
        self.value_offset = lib().mm_ppo_value_offset(h)
        self.obs_dim, self.act_dim = int(obs_dim), int(act_dim)

    @staticmethod
    def _chk(rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed (rc={rc}): {lib().mm_ppo_last_error().decode()}")

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:
            _lib.mm_ppo_destroy(h)

    def _f32(self, t, shape=None):
        assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous() and (shape is None or tuple(t.shape) == tuple(shape)), (t.dtype, t.shape, shape)
        return t.data_ptr()

    def act(self, params, obs, obs_mean, obs_std, noise, obs_out, raw_out, logp_out, value_out, action_out):
        """action_out None: value network only (bootstrap value)"""
        n = obs.shape[0]
        self._f32(params, (self.param_count,)); self._f32(obs, (n, self.obs_dim)); self._f32(value_out, (n,))
        if action_out is not None:
            for t in (noise, raw_out, action_out):
                self._f32(t, (n, self.act_dim))
            self._f32(logp_out, (n,))
            if obs_out is not None:
                self._f32(obs_out, (n, self.obs_dim))
        self._chk(lib().mm_ppo_act(self.h, _ptr(params), _ptr(obs), _ptr(obs_mean), _ptr(obs_std), _ptr(noise) if action_out is not None else None,
                                   int(n), _ptr(obs_out) if action_out is not None else None, _ptr(raw_out) if action_out is not None else None,
                                   _ptr(logp_out) if action_out is not None else None, _ptr(value_out), _ptr(action_out), _stream(obs.device)),
                  "mm_ppo_act")

    def store(self, rwd, rwd_col, reward_scale, ended, truncated, reward_out, trunc_out, term_out):
        n = rwd.shape[0]
        assert ended.dtype == torch.uint8 and (truncated is None or truncated.dtype == torch.uint8)
        self._f32(rwd); self._f32(reward_out, (n,)); self._f32(trunc_out, (n,)); self._f32(term_out, (n,))
        self._chk(lib().mm_ppo_store(_ptr(rwd), int(rwd.shape[1]), int(rwd_col), C.c_float(reward_scale), _ptr(ended), _ptr(truncated), int(n),
                                     _ptr(reward_out), _ptr(trunc_out), _ptr(term_out), _stream(rwd.device)), "mm_ppo_store")

    def grad(self, params, obs, obs_mean, obs_std, idx, raw, logp_old, adv, ret, grad_out):
        """obs [B, obs_dim], raw [B, act_dim], logp_old / adv / ret [B] (the flattened unroll buffers); idx int64 [mb]"""
        B = obs.shape[0]
        assert idx.dtype == torch.int64 and idx.is_cuda and idx.is_contiguous()
        self._f32(params, (self.param_count,)); self._f32(obs, (B, self.obs_dim)); self._f32(raw, (B, self.act_dim))
        for t in (logp_old, adv, ret):
            self._f32(t, (B,))
        self._f32(grad_out, (self.param_count,))
        self._chk(lib().mm_ppo_grad(self.h, _ptr(params), _ptr(obs), _ptr(obs_mean), _ptr(obs_std), idx.data_ptr(), int(idx.numel()), _ptr(raw),
                                    _ptr(logp_old), _ptr(adv), _ptr(ret), _ptr(grad_out), _stream(obs.device)), "mm_ppo_grad")

    def set_entropy_noise(self, noise: Optional[torch.Tensor]):
        """[B, act_dim] standard-normal draws for the entropy's squash log-det-Jacobian sample (brax NormalTanhDistribution.entropy),
        row-indexed like `raw`; None = pre-squash entropy only.  The tensor must stay alive while gradients are taken."""
        if noise is not None:
            self._f32(noise); assert noise.shape[-1] == self.act_dim
        self._ent_noise = noise
        self._chk(lib().mm_ppo_set_entropy_noise(self.h, _ptr(noise)), "mm_ppo_set_entropy_noise")

    def adam(self, params, grad, grad_scale: float = 1.0, recompute_norm: bool = False):
        self._f32(params, (self.param_count,)); self._f32(grad, (self.param_count,))
        self._chk(lib().mm_ppo_adam(self.h, _ptr(params), _ptr(grad), C.c_float(grad_scale), int(bool(recompute_norm)), _stream(params.device)),
                  "mm_ppo_adam")

    def reset_optimizer(self):
        self._chk(lib().mm_ppo_reset_optimizer(self.h, _stream(self.device)), "mm_ppo_reset_optimizer")


def uniform(out: torch.Tensor, seed: int, stream_id: int, first_index: int = 0):
    """out[...] = U[0,1) float32, Philox4x32-10: flat element i is word (first_index+i)%4 of counter ((first_index+i)/4,
    stream_id), key=seed.  A shard of envs passes first_index = env_index_base * nu to draw its slice of the global matrix."""
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
    _chk(lib().mm_uniform_at(out.data_ptr(), out.numel(), C.c_uint64(seed), C.c_uint64(stream_id), int(first_index),
                             _stream(out.device)), "mm_uniform")
    return out


def debug_dump(model: HipModel, state: BatchState, ctrl: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Run mm_forward and return the per-env debug record [E, words] (tests only; fields via model.layout())."""
    total = model.layout("total")
    buf = torch.zeros(state.nenv, total, dtype=torch.float32, device=model.device)
    lib().mm_debug_set_dump(buf.data_ptr())
    try:
        forward(model, state, ctrl)
        torch.cuda.synchronize()
    finally:
        lib().mm_debug_set_dump(None)
    return buf


PROF_STAGES = ["kin", "com", "tendon", "constr", "vel", "crb", "factor", "act", "solve0", "newton", "euler", "io", "total",
               "n_warm", "n_grad", "n_hbuild", "n_factor", "n_solve", "n_prod", "n_ls"]   # the last seven: inside the general-row Newton solve


def profile_stages(fn):
    """Run fn() with in-kernel stage timers on; returns {stage: cycles} of wave 0 / block 0 (tests/tools only)."""
    buf = torch.zeros(2 * len(PROF_STAGES), dtype=torch.int64, device="cuda")     # [main wave | helper wave of a two-wave launch]
    lib().mm_debug_set_prof(buf.data_ptr())
    try:
        fn()
        torch.cuda.synchronize()
    finally:
        lib().mm_debug_set_prof(None)
    v = buf.cpu().tolist()
    out = dict(zip(PROF_STAGES, v[:len(PROF_STAGES)]))
    out.update({"h_" + k: x for k, x in zip(PROF_STAGES, v[len(PROF_STAGES):]) if x})
    return out
