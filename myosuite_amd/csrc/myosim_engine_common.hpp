#pragma once
// myosim_engine_kernel.hpp -- MI355X (gfx950 / CDNA4) batched musculoskeletal physics step, engine v2: device code.
// (The fused kernel template; explicit instantiations live in myosim_inst_*.hip so that they compile in parallel, the
// host side of the C ABI in myosim_engine.hip.)
//
// Execution model ("lane = item"): every environment is owned by a GROUP of G adjacent lanes of one
// 64-wide wavefront (G in {8,16,32,64}; 64/G envs per wave).  Inside the group each lane permanently
// OWNS one item of every kind -- lane g is body g, dof g, joint-limit row g (lower) / g-G/2 (upper) --
// and keeps that item's data in REGISTERS for the whole fused env-step (frame_skip substeps + final
// forward + obs/reward).  Variable-length work (tendon paths, actuators) is swept with lane-strided
// loops.  Only data that other lanes must gather lives in LDS (pose / cdof / composite-inertia tables,
// sparse tendon Jacobian, a dense nv x nv scratch tile); HBM is touched once to load state+action and
// once to store state+obs+reward.
//
// Linear algebra is DENSE and register resident: lane i holds row i of M / H / L.  Cholesky, the two
// triangular solves and M*x run as fully unrolled lane-parallel loops whose only communication is a
// cross-lane broadcast (v_readlane for G = 64, ds_bpermute otherwise): no LDS round trips, no level
// synchronisation.  The constraint Newton solver keeps one (potential) joint-limit row per lane, so no
// compaction is needed.  A wavefront executes in lock-step and the LDS services one wave's
// instructions in order, so stage boundaries need only a compiler fence (GSYNC), never s_barrier.
//
// Pipeline restated (stage order of mj_step; reference call site
// myosuite/robot/robot.py:856-861): kinematics -> comPos -> tendon(+wrap) -> limit rows -> comVel/RNE
// -> CRB -> Cholesky -> passive/actuation -> Newton -> semi-implicit Euler (implicit joint damping).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>

#include "../../include/myosim_model.h"
#include "../../include/myosim.h"

// (MINVALF, the floor of every guarded division, is defined per scalar type in myosim_engine_body.inc)
#ifndef MM_MFMA_HBUILD
#define MM_MFMA_HBUILD 1   /* Newton Hessian update J'DJ of one-env-per-wave kernels on the matrix cores (0: the row-broadcast loop) */
#endif

// ---- Philox4x32-10 (counter based; the oracle side reproduces it in numpy: oracle/env_oracle.py) -----------
__device__ __host__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __host__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ------------------------------------------------------------------ kernel args
struct Dims {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, neq, npair, nM, nlevel, njmax, ntenJ;
  int iterations, ls_iterations, eulerdamp, any_damping;
  int gen;   // model has equality / friction-loss / contact rows: general (dense-J) constraint path
  int nfric; // dofs with frictionloss > 0 (one friction-loss row each, behind the equalities)
  int ntlim; // limited tendons (at most one limit row each, behind the joint-limit rows)
  int dof_nlevel;   // levels of the dof tree (1 + maximum number of ancestor dofs)
  // SP kernels: the dof tree cut into segments (maximal unbranched chains); one lane eliminates a whole segment
  int bchain_nlevel;    // body chains (Engine::subtree_sum): levels of the chain tree | most child chains << 4 | longest chain << 8; 0: the host could not build the chains
  int seg_nlevel;       // levels of the segment tree
  int seg_lvinfo[2];    // one byte per segment level: [3:0] most child segments of a segment there
  int seg_lvtb[2];      // one byte per segment level: [3:0] top depth, [7:4] bottom depth of the segments there (all alike)
  int seg_zero;         // index of the all-zero update-matrix slot (absent children)
  int seg_u;            // word offset (in the u1 LDS region) of the update matrices, 36 words per segment
  int desc_words;       // words of the per-dof descendant list (4 ids each) a product M x has to walk
  int integrator;   // MM_INT_EULER | MM_INT_RK4 | MM_INT_IMPLICITFAST
  int efc_rows;     // allocated rows of the efc_J LDS table: min(lanes_per_env, njmax rounded up to 4)
  int condim4;      // 1: some contact pair is condim 4 (torsional friction: six pyramid rows) -- the torsional pass runs
  int nconmax;      // MM_OI_NCONMAX: contacts beyond this many (in collider order) are dropped and flagged, as mjModel.nconmax
  float timestep, gx, gy, gz, tolerance, ls_tolerance, meaninertia;
  // Origin of the kernel's internal world frame (host: mean body position at qpos0, rounded to 1/64 m).  Physics is
  // translation invariant; fp32 rounding is not: a hand that sits 1 m from the world origin carries ~1e-7 m of absolute
  // error in every point, i.e. ~2e-5 of a 5 mm tendon moment arm.  All positions inside the kernel are relative to this
  // origin; qpos of free joints, task targets and every position OUTPUT stay in world coordinates.
  float ox, oy, oz;
};

// per-env LDS tables (offsets in 32-bit words from the env's base)
struct Layout {
  int qpos, qvel, act, ctrl, actdot;
  int xpos, xmat, xanchor, xaxis, com, cdof;
  int u1;   // union: xquat[4nb] during FK | (cvel,cacc)[12nb] then cfrc[6nb] during the velocity stage | dense NVP*NVP tile afterwards
  int crb;
  int tenlen, tenvel, tenj, tenfrc, actlen, actvel, actfrc;
  int mtile;   // two-wave launches: a second dense NVP x NVP tile (M for the helper wave, which leaves Euler's factor in it) + NVP words (1 / diagonal)
  int flags;   // two-wave launches: [0] passes the main wave has opened (kinematics done), [1] passes the helper wave has finished
  int wrapw;   // per wrapping path item: the two tangent points and a wrapped flag (7 words); inside u1 (free between FK and the velocity stage) when it fits
  int vec;  // nv: joint-transmission actuator forces
  int xvec; // NVP (16-byte aligned): operand vector of M x products routed through LDS
  int rk_qpos0, rk_act0, rk_adot;   // RK4: state at the start of the step, weighted act_dot sum (RK4 models only)
  int tenw, dofw;   // implicitfast: velocity-derivative weights per tendon (b_t - sum_a s_a gear_a^2) and per dof (damping - joint actuators)
  int efcJ, rowtab;   // general constraint rows: J [G][NVP+4] (16-byte aligned rows), row table [G][3] (GEN models only)
  int total;
};

#ifndef MM_FUSE_TENVEL
/* 1: tendon velocities J qvel summed entry by entry inside tendon()'s Jacobian-entry sweep (one LDS float add per entry) instead of a pass
   of their own with one lane per tendon.  Built and measured in round 6 (same-session A/B): hand 7.93 -> 7.70 M env-steps/s, leg 2.00 ->
   1.96 M, self-contact hand 3.65 -> 3.60 M, elbow 32.9 -> 33.4 M -- gfx950 executes a ds_add_f32 at ~3 cycles per active lane, CU-wide
   (tools/micro/lds_atomic_bench.hip), which costs more than the per-tendon walk it replaces.  OFF. */
#define MM_FUSE_TENVEL 0
#endif
#ifndef MM_FOLD_RIGID_SEGMENTS
#define MM_FOLD_RIGID_SEGMENTS 1   /* 0: A/B switch: every site-site segment of a tendon path is a path item of the kernel's sweep (rounds 1-5) */
#endif
// debug dump layout (tests only): one record per env in global memory
struct DbgLayout {
  int xpos, xquat, xipos, cdof, cvel, tenlen, tenvel, tenj, actfrc, actdot, M, bias, smooth, qaccsm, qacc, qfrccon,
      efc_active, efc_D, efc_aref, scal, total;
};

// engine-private tables appended behind the model blob on the device
struct Aux {
  int body_depth, body_rootslot, dof_rootslot;
  int root_list, nroot;
  int jent, jrec;        // tendon Jacobian by entry: [ntenJ][4] {entry, joint word, first record, records}, records [..][4] (host: mm_model_create)
  int item_tab, nitem;   // flattened tendon path items (4 words each), wraps first: see tendon()
  int dof_rel;           // per dof: 64-bit mask (2 words) of the dofs on its kinematic chain (ancestors, descendants, itself)
  int body_dofmask;      // per body: 64-bit mask (2 words) of the dofs between the body and the root of its tree (its chain)
  int dof_desc;          // per dof: ids of all its descendants, one byte each, 0xff-padded to 8 words
  int dof_seg;           // per dof, 6 words: segment owned by the dof's lane (the segment's top dof) or -1; path and child bytes; the dof's depth
  int dof_anc;           // per dof, 2 words: ids of its ancestor dofs by depth, one byte each
  int jnt_pack;          // per joint, 2 words: type | dofadr << 4 | qposadr << 14, bits(qpos0[qposadr]) -- one load instead of type -> address -> qpos0
  int body_chain;        // per body, 3 words: chain owned by the body's lane (its top body): bottom | level << 8 | children << 12, or -1; child chain tops, one byte each
  int ten_len0, ten_len0_f64;   // per tendon: the summed length of its path segments between rigidly connected bodies (folded at create): float table, double table
  int jent_td;           // (MM_FUSE_TENVEL builds only) per tendon-Jacobian entry in processing order: tendon | dof << 16
};

// model constants the kernel reads through the scalar cache (appended to the device blob at KArgs::cofs, see KD / KL / KX)
struct ConstBlock { Dims d; Layout L; Aux x; };

struct KArgs {
  const uint32_t* blob;
  int cofs;              // word offset of the ConstBlock in the device blob
  int sec[MM_NSEC];      // host-side copies (the kernel reads the blob header / ConstBlock instead)
  Dims d;
  Layout L;
  DbgLayout D;
  Aux x;
  mm_state s;
  const float* ctrl;
  mm_task t;
  mm_derived o;
  mm_rollout ro;         // rollout bookkeeping folded into the launch (mm_rollout_step); has_ro = 0: plain mm_env_step
  int has_ro;
  int two_wave;          // every env is run by two waves of the block (Engine::TW): see k_engine
  int has_derived;
  int mode;              // 0: step(s) only, 1: forward only, 2: env step
  float* dbg;
  int blob_words;
  int state_f64;         // precision-mode kernels only (mm64::k_engine): mm_state.qpos / qvel / act / qacc_warmstart point to fp64 rows
                         // (sits in what was padding ahead of `prof`: the fp32 kernels' argument offsets are unchanged)
  unsigned long long* prof;
};
// Stage boundaries as scheduling fences: the machine scheduler works on basic blocks, and with the stage timers compiled out a
// whole forward pass is a handful of very long blocks across which it hoists loads and lengthens live ranges until the 256-VGPR
// kernels spill.  (Found because the tools build, whose timers end a block at every stage, ran the leg kernels 8-15 % FASTER.)
#ifndef MM_STAGE_FENCE
#define MM_STAGE_FENCE 1
#endif
#if MM_STAGE_FENCE
#define MM_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MM_FENCE() ((void)0)
#endif
enum { PF_KIN = 0, PF_COM, PF_TENDON, PF_CONSTR, PF_VEL, PF_CRB, PF_FACTOR, PF_ACT, PF_SOLVE0, PF_NEWTON, PF_EULER,
       PF_IO, PF_TOTAL,
       PF_N_WARM, PF_N_GRAD, PF_N_HBUILD, PF_N_FACTOR, PF_N_SOLVE, PF_N_PROD, PF_N_LS,   // inside the general-row Newton solve (tools build)
       NPROF };

// section offsets come from the blob header in global memory through the scalar cache (s_load at use) instead of ~100
// kernel-argument words that live in (spilled) SGPRs for the whole kernel
typedef const __attribute__((address_space(4))) uint32_t* ConstWords;
#define SECOFF_G_(S) ((int)(reinterpret_cast<ConstWords>(reinterpret_cast<uintptr_t>(a.blob))[MM_HEADER_WORDS + 2 * (MM_SEC_##S)]))
typedef const __attribute__((address_space(4))) ConstBlock ConstBlockC;
typedef const __attribute__((address_space(4))) Dims ConstDims;
typedef const __attribute__((address_space(4))) Layout ConstLayout;
typedef const __attribute__((address_space(4))) Aux ConstAux;
// per-call arguments used late in the kernel (task description, state / derived pointers) are read from the kernarg segment
// at the point of use instead of living in SGPRs from kernel entry
typedef const __attribute__((address_space(4))) KArgs ConstKArgs;
#define KA() (*(ConstKArgs*)(__builtin_amdgcn_kernarg_segment_ptr()))
#define KCB_() (*reinterpret_cast<ConstBlockC*>(reinterpret_cast<uintptr_t>(a.blob + a.cofs)))
// MM_CONST_IN_REGS = 1 (experiment, not the default): the ConstBlock and the section-offset table are read ONCE at kernel entry
// into a by-value struct instead of through the scalar cache at every use (285 s_load per forward pass of the hand kernel,
// SQ_INSTS_SMEM, each followed by an s_waitcnt lgkmcnt(0) that also drains the wave's LDS queue).  Measured on MI355X (A/B in
// one session, tools/gpu_ab.sh): the ~150 extra long-lived wave-uniform values push SGPR spills from 324 to 478 lanes, the
// two extra spill VGPRs tip the 241-VGPR hand kernel into 79 VGPR spills / 296 B scratch, and it LOSES: hand 4.63 -> 4.34 M,
// reorient 2.08 -> 1.73 M env-steps/s, elbow unchanged (its time is dependent-latency, not scalar loads).  With only the
// ConstBlock by value (MM_SEC_IN_REGS = 0): 38 VGPR spills, hand 4.56 M.  The scalar-cache path stays.
#ifndef MM_CONST_IN_REGS
#define MM_CONST_IN_REGS 0
#endif
struct KConst { Dims d; Layout L; Aux x; int sec[MM_NSEC]; };
#ifndef MM_SEC_IN_REGS
#define MM_SEC_IN_REGS 1
#endif
#if MM_CONST_IN_REGS
#if MM_SEC_IN_REGS
#define SECOFF_(S) (kc.sec[MM_SEC_##S])
#else
#define SECOFF_(S) SECOFF_G_(S)
#endif
#define KD() (kc.d)
#define KL() (kc.L)
#define KX() (kc.x)
#else
#define SECOFF_(S) SECOFF_G_(S)
#define KD() (KCB_().d)
#define KL() (KCB_().L)
#define KX() (KCB_().x)
#endif
// A model table = (base of the model words, 32-bit word offset).  Element access builds the BYTE offset in 32 bits and adds it to
// the base as an unsigned value: with the model read through L2 (LM = 0 kernels: `mb` is a uniform global pointer) that is the
// `global_load v, v_off, s[base]` form -- one VGPR and one shift per load -- where indexing a `const T*` with an int index is a
// sign extension + 64-bit add into a VGPR pair per load (659 such loads in the reorient kernel, 7 % of its VALU instructions and
// most of its spills).  Converts to a plain pointer where a callee wants one (the old, slower path).
template <class T>
struct Tab {
  const uint32_t* b;
  uint32_t o;
  __device__ __forceinline__ T operator[](int i) const {
    const uint32_t byte = (o << 2) + (uint32_t)i * (uint32_t)sizeof(T);
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(b) + byte);
  }
  __device__ __forceinline__ Tab operator+(int i) const { return Tab{b, o + (uint32_t)i * (uint32_t)(sizeof(T) / 4)}; }
  __device__ __forceinline__ operator const T*() const { return reinterpret_cast<const T*>(b + o); }
};
#define MI_(S) (Tab<int>{mb, (uint32_t)SECOFF_(S)})
#define MF_(S) (Tab<float>{mb, (uint32_t)SECOFF_(S)})
#define AUXI(f) (Tab<int>{mb, (uint32_t)KX().f})

#define GSYNC()                                           \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

