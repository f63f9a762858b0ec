/* myosim.h -- C ABI of libmyosim_hip.so: the MI355X-native batched physics step.
 *
 * This is the drop-in boundary for the ONE hot path of MyoSuite (SURVEY.md 8b,
 * boundary A): everything the reference does between `env.step(a)` entering
 * BaseV0.step and the observation/reward leaving MujocoEnv.forward, for E
 * independent environments resident in HBM.
 *
 * Reference interfaces each entry point replaces (paths under /root/reference):
 *   mm_model_create   <- MjSpec.from_file(path).compile()      myosuite/envs/env_base.py:72,96-106
 *   mm_env_step       <- BaseV0.step + Robot.step + mj_step loop + MujocoEnv.forward
 *                        myosuite/envs/myo/base_v0.py:82-118, myosuite/robot/robot.py:856-933,
 *                        myosuite/envs/env_base.py:409-459, task get_obs_dict/get_reward_dict
 *                        (myosuite/envs/myo/myobase/pose_v0.py:100-140)
 *   mm_forward        <- mujoco.mj_forward via Robot.sensor2sim  myosuite/robot/robot.py:595-607
 *   mm_env_reset      <- PoseEnvV0.reset / MujocoEnv.reset / Robot.reset (mj_resetData)
 *                        myosuite/envs/myo/myobase/pose_v0.py:174-257, myosuite/robot/robot.py:936-1021
 *   mm_fatigue_*      <- CumulativeFatigue.compute_act/reset     myosuite/envs/myo/fatigue.py:38-99
 *   mm_uniform        <- jax.random.uniform action sampling      benchmarks/mjx_benchmark.py:29
 *
 * Conventions
 *  - plain C, no torch types: every buffer is a raw DEVICE pointer owned by the
 *    caller (PyTorch-ROCm tensors via data_ptr(), or hipMalloc).
 *  - per-env arrays are ENV-MAJOR, row-major [nenv][n] float32: the wave-cooperative
 *    kernels give each env a group of G lanes that sweep the components of that
 *    env, so component-contiguous rows are the coalesced layout.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  No call
 *    synchronises the host; all work is enqueued on `stream`.
 *  - every function returns 0 on success, a negative MM_E* code otherwise, never
 *    throws, never falls back to a CPU path.
 */
#ifndef MYOSIM_H_
#define MYOSIM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM_OK            0
#define MM_EBADBLOB     -1
#define MM_EHIP         -2   /* a HIP runtime call failed: see mm_last_error()   */
#define MM_EUNSUPPORTED -3   /* model uses a feature the engine does not implement */
#define MM_ELDS         -4   /* per-env workspace does not fit in LDS             */
#define MM_EARG         -5

typedef struct mm_model mm_model;

/* task ids for the fused obs/reward stage */
enum { MM_TASK_NONE = 0, MM_TASK_POSE = 1, MM_TASK_REACH = 2, MM_TASK_REORIENT = 3, MM_TASK_WALK = 4, MM_TASK_OBJHOLD = 5,
       MM_TASK_KEYTURN = 6 };

/* mm_model_info selectors */
enum { MM_INFO_NQ = 0, MM_INFO_NV, MM_INFO_NU, MM_INFO_NA, MM_INFO_NBODY, MM_INFO_NSITE, MM_INFO_NTENDON,
       MM_INFO_LANES_PER_ENV, MM_INFO_LDS_BYTES_PER_ENV, MM_INFO_ENVS_PER_BLOCK, MM_INFO_NGEOM,
       MM_INFO_WAVES_PER_BLOCK,
       MM_INFO_KERNEL_FAMILY,   /* 0 limit rows only, dense Cholesky (nv <= 4); 1 limit rows only, tree-sparse L'DL; 2 general rows */
       MM_INFO_MODEL_WORDS,     /* 32-bit words of the device model tables a block stages into LDS when they fit next to its envs */
       MM_INFO_BODY_CHAINS,     /* levels of the body-chain tree | most child chains << 4 | longest chain << 8 (0: level-by-level sweeps) */
       MM_INFO_FOLDED_RESET,    /* 1: mm_rollout.autoreset is available for the WALK / REORIENT tasks on this model (64 lanes per env, a kernel of MM_KERNELS_OBS) */
       MM_INFO_FWD_CARRY,       /* 1: mm_task.fwd_carry is implemented for this model (see mm_task) */
       MM_INFO_TENDON_ITEMS,    /* tendon path items (site-site segments, wraps, fixed-tendon terms) the kernel sweeps per forward pass */
       MM_INFO_TENDON_FOLDED }; /* path segments between two sites whose bodies no dof separates: constant length, summed into the tendon's
                                   constant at mm_model_create instead of being swept (a per-env mm_state.body_pos_env on a body such a
                                   segment spans is refused with MM_EUNSUPPORTED) */

/* ABI version of this header: bumped whenever a struct below gains / loses / reorders a field, an entry point changes its
 * signature or a status / enum value is renumbered.  A caller compares MM_ABI_VERSION (what it was built against) with
 * mm_abi_version() (what the library was built from) and, for bindings that restate the structs (ctypes, cgo, JNI), its own
 * struct sizes with mm_struct_size().  History: 1 = round 1; 2 = mm_state.env_index_base, mm_env_draw(env_index_base), status
  * bits renumbered, the mm_rollout struct -- round 2, shipped under the version STRING of round 1; 3 = this constant + mm_abi_version /
 * mm_struct_size; 4 = mm_task.size / mm_rollout.size (append-only growth of the two structs that gain fields per task); 5 = mm_rollout gains the
 * walk / reorient reset fields (appended: an ABI-4 caller's shorter struct is still accepted); 6 = the "precision" option: under
 * MM_PREC_F64_STATE four mm_state pointers address fp64 rows (no struct changed: an ABI-5 caller that never sets the option is unaffected);
 * 7 = mm_task.fwd_carry appended, mm_gae, mm_model_launch_info, MM_INFO_FWD_CARRY. */
#define MM_ABI_VERSION 7
enum { MM_STRUCT_STATE = 0, MM_STRUCT_DERIVED, MM_STRUCT_TASK, MM_STRUCT_ROLLOUT };

/* Simulation state of a batch, all [nenv][n] float32 device arrays. */
typedef struct {
  int    nenv;            /* >= 1: every call that takes an mm_state refuses an empty batch (MM_EARG), nothing is launched */
  float* qpos;            /* [nenv][nq]                                   */
  float* qvel;            /* [nenv][nv]                                   */
  float* act;             /* [nenv][na]                                   */
  float* qacc_warmstart;  /* [nenv][nv]                                   */
  float* time;            /* [nenv]                                       */
  int32_t* status;        /* [nenv] sticky bits, cleared by reset: 1 bad-state auto reset (mj_step's mj_checkPos / checkVel / checkAcc),
                                    4 the solver hit its iteration cap, 8 contacts or constraint rows were dropped: contacts beyond
                                    mjModel.nconmax (in collider order), or a contact whose rows do not all fit njmax / the engine's 64
                                    rows per env (MuJoCo raises mjWARN_CONTACTFULL / mjWARN_CNSTRFULL for the two; the oracle reports them
                                    as its warn bits 4 / 2 -- tests map both onto this bit), 16 a two-wave launch lost a partner wave (a bounded wait gave up: engine bug),
                                    32 a NaN / Inf / > 1e10 entry in the env's control vector: ALL its controls were set to 0 for this
                                    launch (mj_fwdActuation's mjWARN_BADCTRL); the state is not reset.  After a bad-state reset (bit 1)
                                    the remaining substeps of the launch run on zero controls, as mj_resetData clears them */
  /* per-env model delta (the reference mutates mjModel at reset: reorient_sar_v0.py:407-409): size of ONE geom */
  const float* geom_size_env; /* [nenv][3] or NULL: replaces geom_size[geom_env_id] in collision           */
  int    geom_env_id;         /* geom id the per-env size applies to (-1 = none)                           */
  const int32_t* geom_type_env; /* [nenv] or NULL: replaces geom_type[geom_env_id] (MM_GEOM_CAPSULE / ELLIPSOID /
                                   CYLINDER / BOX; reorient_sar_v0.py:408); pairs with that geom must be authored
                                   against capsules                                                          */
  /* per-env model deltas on ONE body each (-1 / NULL = none): mass (pose_v0.py:177-184 `body_mass[carry_weight]`; the
     inertia tensor is left alone, as the reference leaves it) and frame position in the parent (key_turn_v0.py:163-166
     `body_pos[-1]`) */
  const float* body_mass_env; /* [nenv]    */
  int    body_mass_env_id;
  const float* body_pos_env;  /* [nenv][3] */
  int    body_pos_env_id;
  /* global index of env 0 of this shard (multi-GPU env sharding, myosuite_amd/dist.py): every Philox stream of the reset
     kernels, mm_env_draw and the in-kernel action draw of mm_rollout_step is keyed by env_index_base + e, so a rollout
     does not depend on how the envs are spread over ranks */
  int    env_index_base;
} mm_state;

/* Optional derived outputs of the final forward pass (NULL = not requested). */
typedef struct {
  float* xpos;              /* [nenv][nbody][3] body frame origins, world    */
  float* xquat;             /* [nenv][nbody][4]                               */
  float* xipos;             /* [nenv][nbody][3] body COMs                     */
  float* site_xpos;         /* [nenv][nsite][3]                               */
  float* geom_xpos;         /* [nenv][ngeom][3]                               */
  float* cvel;              /* [nenv][nbody][6] com-frame velocities (rot:lin)*/
  float* subtree_com;       /* [nenv][nbody][3] COM of the kinematic tree the body belongs to (the point cdof / cinert refer to): mjData.subtree_com at tree roots, the ROOT's value in every other slot */
  float* actuator_length;   /* [nenv][nu]                                     */
  float* actuator_velocity; /* [nenv][nu]                                     */
  float* actuator_force;    /* [nenv][nu]                                     */
  float* qacc;              /* [nenv][nv]                                     */
  float* ten_length;        /* [nenv][ntendon]                                */
  int32_t* nefc;            /* [nenv] active constraint rows                  */
  int32_t* solver_niter;    /* [nenv] Newton iterations of the last solve     */
} mm_derived;

/* Per-call description of the env-level (MyoBase) work fused around the physics.
 * Forward compatibility: `size` = sizeof(mm_task) AS THE CALLER WAS COMPILED (first field, always set).  New tasks append
 * their parameters at the END of the struct; the library copies min(size, its own sizeof) bytes and zero-fills the rest, so a
 * caller built against an older, shorter struct keeps working (zero = "feature off" for every appended field), and a caller
 * built against a NEWER header is refused (MM_EARG) instead of having its tail silently ignored. */
typedef struct {
  uint32_t size;            /* sizeof(mm_task) in the caller's build; refused below the ABI-4 struct (everything up to `obs_only`) */
  int   task;               /* MM_TASK_*                                       */
  int   nsubsteps;          /* frame_skip                                      */
  int   normalize_act;      /* 1: muscle ctrl = 1/(1+exp(-5(a-0.5)))  (base_v0.py:86-90) */
  int   do_forward;         /* 1: run the post-step forward pass (sensor2sim)  */
  int   fatigue;            /* 1: 3CC-r fatigue remaps muscle ctrl (fatigue.py) */
  int   max_episode_steps;  /* TimeLimit horizon; 0 = none                     */
  /* POSE task */
  float pose_thd;           /* pose_v0.py:57                                   */
  float far_th;             /* 4*pi/2, pose_v0.py:118                          */
  float w_pose, w_bonus, w_act_reg, w_penalty;   /* pose_v0.py:18-23           */
  const float* target_jnt_value; /* [nenv][nq]                                 */
  /* fatigue state, [nenv][na] each (NULL unless fatigue=1) */
  float* fat_MA; float* fat_MR; float* fat_MF;
  float fat_F, fat_R, fat_r; /* fatigue.py:9-11                                */
  /* outputs */
  float* obs;               /* [nenv][obs_dim] float32, key order of the task  */
  int   obs_dim;
  float* rwd;               /* [nenv][MM_RWD_COUNT] (MM_RWDW_COUNT for WALK): task reward terms */
  uint8_t* done;            /* [nenv]                                          */
  uint8_t* truncated;       /* [nenv] step_count >= max_episode_steps          */
  int32_t* step_count;      /* [nenv] in/out                                   */
  float* ctrl_out;          /* [nenv][nu] optional: ctrl actually applied      */
  int   reaf_src, reaf_dst; /* tendon transfer: ctrl[dst]=ctrl[src]; ctrl[src]=0 (base_v0.py:104-108); -1 = off */
  int   obs_layout;         /* 0: myobase pose  [qpos, qvel*dt, pose_err, act]        (pose_v0.py:17,100-111)
                               1: MJX pose      [qpos, qvel*timestep, act, pose_err]  (playground_pose_v0.py:119-129);
                                  MJX reach     [qpos, qvel*timestep, act, tip_pos, reach_err] (playground_reach_v0.py:150-165) */
  int   act_reg_mean;       /* 1: act_mag = ||act||/na (pose_v0.py:115-117); 0: ||act|| (playground_pose_v0.py:63) */
  float obs_dt;             /* scale of the qvel observation: env.dt (pose_v0.py:104) or opt.timestep (MJX) */
  /* REACH task (envs/myo/myobase/reach_v0.py:95-151): obs [qpos, qvel*dt, tip_pos, reach_err, act];
     reward keys reuse the columns of MM_RWD_* with `pose` := `reach` (w_pose is the reach weight) */
  const int32_t* tip_sites; /* [ntip] site ids (device)                        */
  int   ntip;
  const float* target_pos;  /* [nenv][3*ntip] world positions of the *_target sites */
  float reach_far_th;       /* far_th (per tip), reach_v0.py:57,131-135         */
  int   reach_stand;        /* 1: the leg-stand variant of the reach task (walk_v0.py:17-128, `ReachEnvV0` of walk_v0): reach =
                               10 - reach_dist - 10*||qvel*dt||, act_reg = -100*act_mag, near_th = 0.050 per tip */
  /* WALK task (envs/myo/myobase/walk_v0.py:189-480): obs [qpos[2:], qvel*dt, com_vel(2), torso xquat(4),
     feet_heights(2), height(1), feet_rel_positions(6), phase_var(1), muscle_length, muscle_velocity, muscle_force, act];
     reward columns MM_RWDW_* (row stride MM_RWDW_COUNT).  Needs do_forward = 1. */
  int   walk_body[4];       /* body ids: pelvis, torso, talus_l, talus_r        (walk_v0.py:405-436,498-500) */
  int   walk_qadr[6];       /* qpos addresses: hip_flexion_l, hip_flexion_r (walk_v0.py:465), hip_adduction_l,
                               hip_adduction_r, hip_rotation_l, hip_rotation_r (walk_v0.py:301-303) */
  float walk_min_height;    /* walk_v0.py:248,382-388                           */
  float walk_max_rot;       /* walk_v0.py:249,514-526                           */
  int   walk_hip_period;    /* walk_v0.py:250,291,456                           */
  float walk_target_x_vel, walk_target_y_vel;   /* walk_v0.py:252-253,444-451   */
  float walk_target_rot[4]; /* init_qpos[3:7] unless given (walk_v0.py:472-479) */
  float walk_w[5];          /* weights of vel_reward, done, cyclic_hip, ref_rot, joint_angle_rew (walk_v0.py:207-213) */
  /* REORIENT task (envs/myo/myobase/reorient_sar_v0.py:116-174): obs [hand_jnt = qpos[:-6], obj_pos, obj_vel = qvel[-6:]*dt,
     obj_rot, obj_des_rot, obj_err_pos, obj_err_rot, mlen, mvel, mforce, act]; reward columns MM_RWDR_*.  The reference
     moves the top/bot marker geoms to +-axis_half along the object's z at reset (:403-406) but keeps pen_length /
     tar_length from setup (:84-91): obj_rot = R_obj * (0,0,2*axis_half) / pen_length. */
  int   reor_obj_body;      /* "Object" body id                                 */
  int   reor_eps_site;      /* "eps_ball" site id (obj_des_pos)                 */
  float reor_pen_length;    /* |geom_pos[top] - geom_pos[bot]| of the compiled model */
  const float* reor_axis_half; /* [nenv]                                        */
  const float* reor_des_rot;   /* [nenv][3] obj_des_rot of the episode (target body quat applied, / tar_length) */
  float reor_w[5];          /* weights of pos_align, rot_align, act_reg, drop, bonus (reorient_sar_v0.py:38-44) */
  int   reor_obs_muscle;    /* 1: obs carries mlen / mvel / mforce (reorient_sar_v0.py); 0: PenTwirl's obs (pen_v0.py:16-26:
                               hand_jnt, obj_pos, obj_vel, obj_rot, obj_des_rot, obj_err_pos, obj_err_rot, act)          */
  /* OBJHOLD task (envs/myo/myobase/obj_hold_v0.py:60-131; free-joint object = the last 7 qpos / 6 qvel): obs
     [hand_qpos = qpos[:-7], hand_qvel = qvel[:-6]*dt, obj_pos, obj_err = goal - obj_pos, act]; reuses tip_sites[0] (the
     "object" site), target_pos [nenv][3] (the per-episode "goal" site position), w_pose (goal_dist weight), w_bonus,
     w_penalty, w_act_reg and the MM_RWD_* columns (POSE := goal_dist). */
  /* KEYTURN task (envs/myo/myobase/key_turn_v0.py:84-150; the key's hinge is the last qpos / qvel): obs [hand_qpos =
     qpos[:-1], hand_qvel = qvel[:-1]*dt, key_qpos, key_qvel*dt, IFtip_approach = keyhead - IFtip, THtip_approach = keyhead -
     THtip, act]; tip_sites[0..2] = the keyhead, IFtip, THtip site ids; reward columns MM_RWDK_*.  Needs do_forward = 1. */
  float key_goal_th;        /* key_turn_v0.py:59,138                            */
  float key_w[6];           /* weights of key_turn, IFtip_approach, THtip_approach, act_reg, bonus, penalty (key_turn_v0.py:24-31) */
  /* reset observation support (all tasks) */
  const uint8_t* env_mask;  /* optional [nenv]: envs with 0 are left untouched  */
  int   obs_only;           /* 1: no substeps, no ctrl map, no counters, no reward write: forward + obs of the CURRENT state
                               (the observation returned by reset(): env_base.py:560-575) */
  /* --- appended in ABI 7: forward-pass carry, optional [nenv][2 nv + 1] float32, zero-initialised by the caller and otherwise
     owned by the engine.  The trailing mj_forward of env.step k (robot.py:595-607) evaluates exactly the state the first
     mj_step of env.step k + 1 evaluates again (robot.py:856-861) -- the new action only enters act_dot when every actuator has
     activation dynamics -- so the trailing pass leaves (qacc, the Euler step's damped acceleration) here under a hash of the
     state rows and per-env model deltas, and the next launch's first substep starts from them when the hash matches what it
     loaded: one pipeline pass in frame_skip + 1 saved, results bit-identical.  Any change of the state from outside (reset,
     set_state, a tensor write) changes the hash and the row is simply ignored.  MM_INFO_FWD_CARRY says whether the model's
     kernel family implements it (Euler, fp32, more than 4 dofs, no stateless actuators); MM_EUNSUPPORTED otherwise. */
  float* fwd_carry;
} mm_task;

/* Rollout bookkeeping folded into the env-step launch (mm_rollout_step): what a rollout harness around env.step does per
 * step -- benchmarks/mjx_benchmark.py:29 draws actions, gym's autoreset wrapper / playground's TrainingWrapper re-arm
 * finished episodes, RecordEpisodeStatistics accumulates returns -- without extra kernel launches. */
typedef struct {
  uint32_t size;            /* sizeof(mm_rollout) in the caller's build (same rule as mm_task.size; minimum: everything up to `reset_seed`) */
  const float* action;      /* [nenv][nu], or NULL: draw action ~ U[0,1) in the kernel (Philox4x32-10, mm_uniform's scheme:
                               element i = (env_index_base + e) * nu + u is word i%4 of counter (i/4, action_stream), key action_seed) */
  uint64_t action_seed, action_stream;
  float* action_out;        /* optional [nenv][nu]: the actions that were applied (for the learner)                 */
  float* ep_stats;          /* optional [nenv][3] in/out: return += dense reward, length += 1, solved = max(solved, .) */
  uint8_t* reset_mask;      /* optional [nenv] out: done | truncated of this step                                  */
  /* masked auto-reset folded into the same launch, MM_TASK_POSE (pose_v0.py:174-257; same draws as mm_pose_reset; WALK / REORIENT:
     see the fields at the end):
     envs whose episode ended get qpos ~ U(qlo,qhi) (random_qpos) or qpos0, target ~ U(tlo,thi), qvel = act = time = 0,
     step_count = 0, episode += 1, and their obs row holds the FIRST observation of the new episode (reward / done rows keep
     the terminal step's values).  Other tasks: autoreset = 0, reset through reset_mask + the task's reset call. */
  int   autoreset;
  int   random_qpos;
  const float *qlo, *qhi, *tlo, *thi;   /* [nq] each */
  float* target;            /* [nenv][nq]: == mm_task.target_jnt_value                                             */
  int32_t* episode;         /* [nenv] in/out                                                                        */
  uint64_t reset_seed;
  /* --- appended in ABI 5: the masked auto-reset of the WALK and REORIENT tasks folded into the launch as well.  Their first
     observation needs a forward pass on the reset state, so the launch makes a second pass (forward + observation) for the envs
     it re-arms; available where an env is a whole wavefront (64 lanes per env: the leg and reorient models), MM_EUNSUPPORTED
     otherwise.  Same draws, state and per-env model deltas as mm_walk_reset / mm_reorient_reset_typed (episode counter included);
     the 3CC-r state of a re-armed env goes back to rest (MF = fat_reset_vec or 0, MR = 1 - MF, MA = 0: mm_fatigue_reset). */
  const float *walk_ka_qpos, *walk_ka_qvel;   /* [nq] / [nv] key pose (walk_v0.py:354-365); kb_*: second stride key of the "random" reset */
  const float *walk_kb_qpos, *walk_kb_qvel;
  int   walk_random;
  const float* reor_init_qpos;      /* [nq]                                                                         */
  const float* reor_size_tables;    /* [4][reor_ntab][3]: capsule, ellipsoid, cylinder, box                         */
  int   reor_ntab;
  float reor_tar_length;
  float*   reor_geom_size_env;      /* [nenv][3] = mm_state.geom_size_env (written)                                 */
  int32_t* reor_geom_type_env;      /* [nenv]    = mm_state.geom_type_env (written)                                 */
  float*   reor_axis_half;          /* [nenv]    = mm_task.reor_axis_half (written)                                 */
  float*   reor_des_rot;            /* [nenv][3] = mm_task.reor_des_rot (written)                                   */
  const float* fat_reset_vec;       /* [na] or NULL                                                                 */
} mm_rollout;

/* columns of mm_task.rwd for MM_TASK_POSE (pose_v0.py:120-139) */
enum { MM_RWD_POSE = 0, MM_RWD_BONUS, MM_RWD_PENALTY, MM_RWD_ACT_REG, MM_RWD_SPARSE, MM_RWD_SOLVED,
       MM_RWD_DONE, MM_RWD_DENSE, MM_RWD_COUNT };
/* columns of mm_task.rwd for MM_TASK_WALK (walk_v0.py:305-325) */
enum { MM_RWDW_VEL = 0, MM_RWDW_CYCLIC_HIP, MM_RWDW_REF_ROT, MM_RWDW_JOINT_ANGLE, MM_RWDW_ACT_MAG, MM_RWDW_SPARSE,
       MM_RWDW_SOLVED, MM_RWDW_DONE, MM_RWDW_DENSE, MM_RWDW_COUNT };

/* columns of mm_task.rwd for MM_TASK_KEYTURN (key_turn_v0.py:116-150) */
enum { MM_RWDK_KEY_TURN = 0, MM_RWDK_IF_APPROACH, MM_RWDK_TH_APPROACH, MM_RWDK_ACT_REG, MM_RWDK_BONUS, MM_RWDK_PENALTY,
       MM_RWDK_SPARSE, MM_RWDK_SOLVED, MM_RWDK_DONE, MM_RWDK_DENSE, MM_RWDK_COUNT };

/* columns of mm_task.rwd for MM_TASK_REORIENT (reorient_sar_v0.py:136-166) */
enum { MM_RWDR_POS_ALIGN = 0, MM_RWDR_ROT_ALIGN, MM_RWDR_ACT_REG, MM_RWDR_DROP, MM_RWDR_BONUS, MM_RWDR_SPARSE,
       MM_RWDR_SOLVED, MM_RWDR_DONE, MM_RWDR_DENSE, MM_RWDR_COUNT };

/* ---- model ---------------------------------------------------------------- */
int  mm_model_create(const uint32_t* blob_host, int nwords, mm_model** out);
void mm_model_destroy(mm_model* m);
int  mm_model_info(const mm_model* m, int which);
/* lanes_per_env in {4,8,16,32,64}; 0 = engine default for the model size. */
int  mm_model_set_lanes(mm_model* m, int lanes_per_env);
/* tuning knobs: "lds_model" (1 = stage the model tables in LDS unless that costs resident waves the batch needs,
   0 = never, 2 = always), "waves_per_block" (0 = auto), "origin_shift" (1 = the kernel works in a frame centred on the
   model; 0 = raw world coordinates, for the fp32 error study),
   "precision" (MM_PREC_*, below), "iterations" / "ls_iterations" (mjOption.iterations / ls_iterations of this handle; the
   blob's values are the default -- the reference's MJX envs set both to 6 after loading, envs/myo/mjx/mjx_base_env.py:50-51) */
int  mm_model_set_option(mm_model* m, const char* name, int value);
/* "precision": which kernel family steps the model.
     MM_PREC_F32        the default: fp32 arithmetic, tables and state rows (the throughput kernels).
     MM_PREC_F64        the same pipeline with fp64 arithmetic, registers and on-chip tables; every buffer of the ABI keeps its
                        type (state rows fp32: the state is rounded once per launch, not per substep), so any caller can switch.
     MM_PREC_F64_STATE  as MM_PREC_F64, and mm_state.qpos / qvel / act / qacc_warmstart point to FLOAT64 rows [nenv][n] (the
                        pointers keep their declared type; time, actions, targets, observations, rewards stay fp32).  This is
                        the mode that meets "state divergence vs CPU mj_step < 1e-4 rel over 1000 steps" on every env: MuJoCo's
                        mjtNum is double, and an fp32 state row alone already breaks the bound on ~1.5 % of the hand's envs.
   Available for limit-rows-only models (no contacts / equalities / friction loss) on the Euler integrator with nv <= 24 --
   BASELINE.json's configs 2-3; MM_EUNSUPPORTED otherwise.  Model tables (the MYOB blob) are fp32 in every mode.  */
enum { MM_PREC_F32 = 0, MM_PREC_F64 = 1, MM_PREC_F64_STATE = 2 };
/* lanes per env a launch over `nenv` envs will use (the width is picked per launch from the batch size unless pinned
   with mm_model_set_lanes or fixed by the model's constraint tables) */
int  mm_model_launch_lanes(const mm_model* m, int nenv);
/* geometry and occupancy of the env-step launch over `nenv` envs, nothing is launched: out[MM_LAUNCH_*], nout >= MM_LAUNCH_COUNT.
   WAVES_PER_BLOCK counts the helper waves of a two-wave launch; RESIDENT_BLOCKS_PER_CU is the HIP occupancy of that kernel at
   that block size and LDS footprint (what bounds it: LDS bytes, VGPRS); a profile's per-wave counters are priced per resident
   wave with it (bench.py). */
enum { MM_LAUNCH_LANES = 0, MM_LAUNCH_WAVES_PER_BLOCK, MM_LAUNCH_TWO_WAVE, MM_LAUNCH_LDS_MODEL, MM_LAUNCH_LDS_BYTES, MM_LAUNCH_BLOCKS,
       MM_LAUNCH_RESIDENT_BLOCKS_PER_CU, MM_LAUNCH_VGPRS, MM_LAUNCH_COUNT };
int  mm_model_launch_info(const mm_model* m, int nenv, int* out, int nout);

/* ---- physics -------------------------------------------------------------- */
/* `nsub` mj_step substeps with ctrl [nenv][nu] applied as-is (engine boundary). */
int  mm_step(const mm_model* m, const mm_state* s, const float* ctrl, int nsub, void* stream);
/* mj_forward on the current state; fills the requested derived arrays. */
int  mm_forward(const mm_model* m, const mm_state* s, const float* ctrl, const mm_derived* out, void* stream);
/* fused env.step: action [nenv][nu] -> state advanced, obs/reward/done written. */
int  mm_env_step(const mm_model* m, const mm_state* s, const float* action, const mm_task* t,
                 const mm_derived* out, void* stream);

/* mm_env_step with the rollout bookkeeping of `r` folded into the same launch (in-kernel action draw, episode statistics,
 * reset mask and -- POSE task -- the masked auto-reset with the first observation of the new episode). */
int  mm_rollout_step(const mm_model* m, const mm_state* s, const mm_task* t, const mm_rollout* r, const mm_derived* out,
                     void* stream);

/* ---- reset / RNG ----------------------------------------------------------- */
/* mj_resetData + (qpos,qvel) overwrite for envs with mask[e]!=0 (mask NULL = all).
 * qpos_src/qvel_src may be NULL (=> qpos0 / 0) and are [nenv][nq] / [nenv][nv]. */
int  mm_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qpos_src,
              const float* qvel_src, void* stream);
/* Pose-task reset: for masked envs draw qpos ~ U(lo,hi)[nq] and target ~ U(tlo,thi)[nq]
 * from Philox4x32-10 keyed by (seed, env, episode counter), then mj_resetData. */
int  mm_pose_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qlo,
                   const float* qhi, const float* tlo, const float* thi, float* target, int32_t* episode,
                   int32_t* step_count, uint64_t seed, int random_qpos, float* obs, int obs_dim,
                   int obs_layout, void* stream);
/* (obs != NULL: the first observation of the new episode is written for the reset envs) */
/* Reach-task reset (reach_v0.py:153-172): targets ~ U(tlo,thi)[3*ntip] (Philox keyed by seed, env, episode),
 * state = mj_resetData (qpos0); the first observation uses tip0 = tip positions at qpos0. */
int  mm_reach_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* tlo, const float* thi,
                    float* target, const float* tip0, int ntip, int32_t* episode, int32_t* step_count,
                    uint64_t seed, float* obs, int obs_dim, void* stream);
/* Walk-task reset (walk_v0.py:327-365): reset_type 0 -> key_a, 1 ("random") -> key_a or key_b by a Philox coin, plus
 * N(0, 0.02) on every qpos coordinate except root height (qpos[2]) and root quaternion (qpos[3:7]); qvel = the key's
 * qvel; act = 0, time = 0, step_count = 0.  key_*: device [nq] / [nv].  The first observation is produced by
 * mm_env_step with mm_task.obs_only = 1 and env_mask = mask. */
int  mm_walk_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* key_a_qpos,
                   const float* key_a_qvel, const float* key_b_qpos, const float* key_b_qvel, int random,
                   int32_t* episode, int32_t* step_count, uint64_t seed, void* stream);
/* Reorient-task reset (reorient_sar_v0.py:265-437, capsule branch): per env a size row of `size_table` [ntab][3]
 * (Philox index draw) -> geom_size_env[e], axis_half[e] = 1.3 * size[1]; desired orientation
 * euler2quat([U(-1,1), U(-0.8,1.2), 0]) (utils/quat_math.py:70-86) -> des_rot[e] = R * (0,0,2*axis_half) / tar_length;
 * state = init_qpos [nq], qvel = act = 0.  First observation: mm_env_step with obs_only. */
int  mm_reorient_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                       const float* size_table, int ntab, float* geom_size_env, float* axis_half, float* des_rot,
                       float tar_length, int32_t* episode, int32_t* step_count, uint64_t seed, void* stream);
/* Pen-twirl reset (pen_v0.py:171-184): fixed object geometry; des_rot[e] = R(euler2quat([U(lo0,hi0), U(lo1,hi1), 0])) *
 * (0,0,2*axis_half) / tar_length, with lo = hi = 0 for the Fixed task; state = init_qpos, qvel = act = 0. */
int  mm_pen_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos, float axis_half,
                  float lo0, float hi0, float lo1, float hi1, float* des_rot, float tar_length, int32_t* episode,
                  int32_t* step_count, uint64_t seed, void* stream);
/* Same with the object TYPE drawn too (reorient_sar_v0.py:388-406): size_tables [4][ntab][3] in the order capsule,
 * ellipsoid, cylinder, box (geom types 3,4,5,6); geom_type_env[e] receives the type; axis_half = 1.3*size[1] (capsule),
 * size[2] (ellipsoid, box), size[1] (cylinder). */
int  mm_reorient_reset_typed(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                             const float* size_tables, int ntab, float* geom_size_env, int32_t* geom_type_env,
                             float* axis_half, float* des_rot, float tar_length, int32_t* episode, int32_t* step_count,
                             uint64_t seed, void* stream);
/* Object-hold reset (obj_hold_v0.py:134-145): goal[e] = goal_center + U(-goal_half, goal_half)^3, object size[e] ~
 * U(size_lo, size_hi)^3 -> geom_size_env (skipped when goal_half == 0 / geom_size_env == NULL: the Fixed task); state =
 * init_qpos, qvel = act = 0. */
int  mm_objhold_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                      const float* goal_center, float goal_half, float size_lo, float size_hi, float* goal,
                      float* geom_size_env, int32_t* episode, int32_t* step_count, uint64_t seed, void* stream);
/* Per-episode draw of a per-env model delta (pose_v0.py:180-183 weight ~ U(weight_range); key_turn_v0.py:164-166 key
 * position offset): out[e][k] = base[k] + lo[k] + (hi[k]-lo[k]) * u, u = Philox4x32-10 word k%4 of counter
 * (k/4, stream_id, env_index_base + e, episode[e]), key = seed; envs with mask[e] == 0 are left untouched.  Call BEFORE the task reset of the
 * same episode (which increments episode[e]).  base may be NULL (= 0). */
int  mm_env_draw(float* out, int nenv, int ncomp, const float* base, const float* lo, const float* hi, const uint8_t* mask,
                 const int32_t* episode, uint64_t seed, uint32_t stream_id, int env_index_base, void* stream);
/* 3CC-r fatigue state of the masked envs back to rest (CumulativeFatigue.reset, fatigue.py:82-99): MF = fatigue_reset_vec
 * (NULL = 0), MR = 1 - MF, MA = 0; arrays are [nenv][na] (the buffers of mm_task.fat_*). */
int  mm_fatigue_reset(float* MA, float* MR, float* MF, const uint8_t* mask, const float* fatigue_reset_vec, int nenv, int na,
                      void* stream);
/* Rollout bookkeeping in one launch (what a gym vector wrapper's RecordEpisodeStatistics + autoreset mask do with a handful
 * of elementwise ops): stats[e] = {return += rwd[e][dense_col], length += 1, solved = max(solved, rwd[e][solved_col])},
 * reset_mask[e] = done[e] | truncated[e].  stats is [nenv][3] float32, rwd has row stride rwd_cols. */
int  mm_episode_stats(float* stats, uint8_t* reset_mask, const float* rwd, int rwd_cols, int dense_col, int solved_col,
                      const uint8_t* done, const uint8_t* truncated, int nenv, void* stream);
/* Generalised advantage estimation of a T-step unroll, one launch (the learner side of benchmarks/mjx_benchmark_PPO.py:50-60):
 * brax's compute_gae term by term -- delta_t = (r_t + gamma (1 - terminated_t) V_{t+1} - V_t)(1 - truncated_t);
 * acc_t = delta_t + gamma lambda (1 - terminated_t)(1 - truncated_t) acc_{t+1}; returns_t = vs_t = acc_t + V_t;
 * advantage_t = (r_t + gamma (1 - terminated_t) vs_{t+1} - V_t)(1 - truncated_t), vs_T = V_T.  A truncated step neither bootstraps
 * from V_{t+1} (after an auto-reset that is the next episode's first observation) nor carries an advantage.  reward / terminated /
 * truncated (may be NULL) / advantage / returns are [T][nenv] float32, value is [T+1][nenv]. */
int  mm_gae(const float* reward, const float* terminated, const float* truncated, const float* value, float* advantage,
            float* returns, int T, int nenv, float gamma, float lam, void* stream);
/* out[i] = U[0,1) float32 from Philox4x32-10, counter = (i, stream_id), key = seed */
int  mm_uniform(float* out, size_t n, uint64_t seed, uint64_t stream_id, void* stream);
/* the same stream from element `first_index` on: out[i] = element first_index + i (a shard of envs draws ITS slice of the
 * global [nenv_total][nu] action matrix: first_index = env_index_base * nu) */
int  mm_uniform_at(float* out, size_t n, uint64_t seed, uint64_t stream_id, size_t first_index, void* stream);

const char* mm_last_error(void);
const char* mm_version(void);
int  mm_abi_version(void);                 /* MM_ABI_VERSION of the header the library was compiled from */
int  mm_struct_size(int which);            /* sizeof of the MM_STRUCT_* struct as the library sees it (negative: unknown selector) */

#ifdef __cplusplus
}
#endif
#endif /* MYOSIM_H_ */
