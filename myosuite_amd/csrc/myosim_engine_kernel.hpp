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

#define MINVALF 1e-15f
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

// ------------------------------------------------------------------ small math
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { V3 r = {x, y, z}; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 ld3(Tab<float> p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 ldq(const float* p) { Q4 q = {p[0], p[1], p[2], p[3]}; return q; }
__device__ __forceinline__ Q4 ldq(Tab<float> p) { Q4 q = {p[0], p[1], p[2], p[3]}; return q; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  Q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}
__device__ __forceinline__ Q4 qnorm(Q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < MINVALF) { Q4 r = {1.f, 0.f, 0.f, 0.f}; return r; }
  float i = 1.f / n;
  Q4 r = {q.w * i, q.x * i, q.y * i, q.z * i};
  return r;
}
struct M3 { float m[9]; };
__device__ __forceinline__ M3 q2m(Q4 q) {
  M3 r;
  float w = q.w, x = q.x, y = q.y, z = q.z;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[4] = w * w - x * x + y * y - z * z; r.m[8] = w * w - x * x - y * y + z * z;
  r.m[1] = 2.f * (x * y - w * z); r.m[3] = 2.f * (x * y + w * z);
  r.m[2] = 2.f * (x * z + w * y); r.m[6] = 2.f * (x * z - w * y);
  r.m[5] = 2.f * (y * z - w * x); r.m[7] = 2.f * (y * z + w * x);
  return r;
}
__device__ __forceinline__ M3 ldm(const float* p) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.m[i] = p[i];
  return r;
}
__device__ __forceinline__ V3 mv(const M3& m, V3 v) {
  return v3(m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z, m.m[3] * v.x + m.m[4] * v.y + m.m[5] * v.z,
            m.m[6] * v.x + m.m[7] * v.y + m.m[8] * v.z);
}
__device__ __forceinline__ V3 mtv(const M3& m, V3 v) {
  return v3(m.m[0] * v.x + m.m[3] * v.y + m.m[6] * v.z, m.m[1] * v.x + m.m[4] * v.y + m.m[7] * v.z,
            m.m[2] * v.x + m.m[5] * v.y + m.m[8] * v.z);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// sin/cos of a joint half-angle (|x| <= ~8), evaluated in fp64 and rounded ONCE to fp32 (<= 0.5 ulp).  Round 2 used an fp32
// Cody-Waite + minimax form with ~1.5e-7 absolute error: 2.5 ulp on the quaternion of EVERY joint, i.e. ~3e-7 rad of
// orientation error per joint, which a 0.2 m lever arm and a 9-joint chain turn into ~1.5e-7 m of position error -- 3x what
// fp32 storage alone costs, and the largest term in the kernel's tendon moment-arm error (moment arms are ~5 mm, path
// segments as short as that).  v_fma_f64 issues at the fp32 FMA rate on gfx950 and this runs once per joint per pass.
#ifndef MM_SINCOS_F64
#define MM_SINCOS_F64 1
#endif
__device__ __forceinline__ void sincos_small(float x, float* s, float* c) {
#if MM_SINCOS_F64
  const double xd = (double)x;
  const double k = __builtin_rint(xd * 0.63661977236758134308);
  const double r = __builtin_fma(-k, 1.57079632679489661923, xd);   // |k| <= 6: the product is exact to 1e-15
  const double r2 = r * r;
  double sp = -2.5052108385441718775e-08;                            // Taylor to r^11 / r^12 on |r| <= pi/4: error < 1e-11
  sp = __builtin_fma(sp, r2, 2.7557319223985890653e-06);
  sp = __builtin_fma(sp, r2, -1.9841269841269841270e-04);
  sp = __builtin_fma(sp, r2, 8.3333333333333333333e-03);
  sp = __builtin_fma(sp, r2, -1.6666666666666666667e-01);
  const double sn = __builtin_fma(sp * r2, r, r);
  double cp = 2.0876756987868098979e-09;
  cp = __builtin_fma(cp, r2, -2.7557319223985890653e-07);
  cp = __builtin_fma(cp, r2, 2.4801587301587301587e-05);
  cp = __builtin_fma(cp, r2, -1.3888888888888888889e-03);
  cp = __builtin_fma(cp, r2, 4.1666666666666666667e-02);
  cp = __builtin_fma(cp, r2, -0.5);
  const double cs = __builtin_fma(cp, r2, 1.0);
  const int q = (int)k & 3;
  const double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
  *s = (float)((q & 2) ? -ss : ss);
  *c = (float)(((q + 1) & 2) ? -cc : cc);
#else
  float k = rintf(x * 0.636619772367581f);
  float r = fmaf(-k, 1.5707963705062866f, x);
  r = fmaf(-k, -4.371138828673793e-8f, r);
  float r2 = r * r;
  float sp = fmaf(fmaf(fmaf(2.718311493989822e-6f, r2, -1.984090227e-4f), r2, 8.3333169e-3f), r2, -0.16666667f);
  float sn = fmaf(sp * r2, r, r);
  float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625e-3f), r2, 4.166664568e-2f), r2, -0.5f);
  float cs = fmaf(cp, r2, 1.f);
  int q = (int)k & 3;
  float ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
#endif
}

// spatial inertia (Ixx Iyy Izz Ixy Ixz Iyz, m*r[3], m) times motion vector [w; v]
__device__ __forceinline__ void inert_mul(float* res, const float* I, const float* v) {
  V3 w = ld3(v), l = ld3(v + 3), mr = ld3(I + 6);
  V3 c1 = cross(mr, l), c2 = cross(mr, w);
  res[0] = I[0] * w.x + I[3] * w.y + I[4] * w.z + c1.x;
  res[1] = I[3] * w.x + I[1] * w.y + I[5] * w.z + c1.y;
  res[2] = I[4] * w.x + I[5] * w.y + I[2] * w.z + c1.z;
  res[3] = I[9] * l.x - c2.x; res[4] = I[9] * l.y - c2.y; res[5] = I[9] * l.z - c2.z;
}
__device__ __forceinline__ void cross_motion(float* res, const float* v, const float* s) {
  V3 w = ld3(v), l = ld3(v + 3), sa = ld3(s), sl = ld3(s + 3);
  st3(res, cross(w, sa)); st3(res + 3, cross(w, sl) + cross(l, sa));
}
__device__ __forceinline__ void cross_force(float* res, const float* v, const float* f) {
  V3 w = ld3(v), l = ld3(v + 3), fa = ld3(f), fl = ld3(f + 3);
  st3(res, cross(w, fa) + cross(l, fl));
  st3(res + 3, cross(w, fl));
}

// ---------------------------------------------------------------- group helpers
// broadcast lane j (group-uniform index) of the group
template <int G>
__device__ __forceinline__ float bc(float v, int j) {
  const int iv = __builtin_bit_cast(int, v);
  if constexpr (G == 64) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, j));
  } else if constexpr (G == 32) {
    // one v_readlane per env of the wave + a select: no LDS crossbar round trip
    int s0 = __builtin_amdgcn_readlane(iv, j), s1 = __builtin_amdgcn_readlane(iv, j + 32);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? s1 : s0);
  } else if constexpr (G == 16) {
    int s0 = __builtin_amdgcn_readlane(iv, j), s1 = __builtin_amdgcn_readlane(iv, j + 16);
    int s2 = __builtin_amdgcn_readlane(iv, j + 32), s3 = __builtin_amdgcn_readlane(iv, j + 48);
    const int q = (threadIdx.x >> 4) & 3;
    return __builtin_bit_cast(float, q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3)));
  } else {
    return __shfl(v, j, G);
  }
}
// gather from a lane-varying source inside the group
template <int G>
__device__ __forceinline__ float sh(float v, int src) { return __shfl(v, src, G); }

// ---- group reductions on the DPP network (no LDS crossbar round trips) ----------------------------------------
// Stages: xor 1 / xor 2 inside quads (quad_perm), quads -> 8 lanes (row_half_mirror), 8 -> 16 lanes (row_mirror); rows of
// 16 are combined through v_readlane.  Every stage is symmetric (lane i and its partner compute a op b and b op a), so
// the result is BITWISE IDENTICAL in every lane of the group -- group-uniform decisions (line-search alpha, loop exits)
// rely on that.  The reduced value is made opaque first (gsum): a contracted fma(a_i, b_i, partner) would differ between partners.
#define DPP_QUAD_XOR1 0xB1
#define DPP_QUAD_XOR2 0x4E
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ float rl(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int G>
__device__ __forceinline__ float gsum(float v) {
  // the first stage must add the ROUNDED operand: __fadd_rn is a plain `+` in this toolchain, so a product passed in would be
  // contracted to fma(a_i, b_i, partner), which differs from the partner's fma(a_j, b_j, mine) in the last bit -- enough to
  // split a group's line search at a kink of the cost (seen with friction-loss rows: even / odd lanes took different alpha)
  asm("" : "+v"(v));
  v = __fadd_rn(v, dppf<DPP_QUAD_XOR1>(v));
  v = __fadd_rn(v, dppf<DPP_QUAD_XOR2>(v));
  if constexpr (G >= 8) v = __fadd_rn(v, dppf<DPP_ROW_HALF_MIRROR>(v));
  if constexpr (G >= 16) v = __fadd_rn(v, dppf<DPP_ROW_MIRROR>(v));
  if constexpr (G == 32) {
    float a0 = __fadd_rn(rl(v, 0), rl(v, 16)), a1 = __fadd_rn(rl(v, 32), rl(v, 48));
    v = (threadIdx.x & 32) ? a1 : a0;
  }
  if constexpr (G == 64) v = __fadd_rn(__fadd_rn(rl(v, 0), rl(v, 16)), __fadd_rn(rl(v, 32), rl(v, 48)));
  return v;
}
template <int G>
__device__ __forceinline__ float gmax(float v) {
  v = fmaxf(v, dppf<DPP_QUAD_XOR1>(v));
  v = fmaxf(v, dppf<DPP_QUAD_XOR2>(v));
  if constexpr (G >= 8) v = fmaxf(v, dppf<DPP_ROW_HALF_MIRROR>(v));
  if constexpr (G >= 16) v = fmaxf(v, dppf<DPP_ROW_MIRROR>(v));
  if constexpr (G == 32) {
    float a0 = fmaxf(rl(v, 0), rl(v, 16)), a1 = fmaxf(rl(v, 32), rl(v, 48));
    v = (threadIdx.x & 32) ? a1 : a0;
  }
  if constexpr (G == 64) v = fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
  return v;
}
template <int G>
__device__ __forceinline__ int gor(int v) {
  v |= dppi<DPP_QUAD_XOR1>(v);
  v |= dppi<DPP_QUAD_XOR2>(v);
  if constexpr (G >= 8) v |= dppi<DPP_ROW_HALF_MIRROR>(v);
  if constexpr (G >= 16) v |= dppi<DPP_ROW_MIRROR>(v);
  if constexpr (G == 32) {
    int a0 = __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16);
    int a1 = __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
    v = (threadIdx.x & 32) ? a1 : a0;
  }
  if constexpr (G == 64)
    v = __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
  return v;
}

// ------------------------------------------------------------- tendon wrapping (A2)
__device__ __forceinline__ bool seg_intersect(float p1x, float p1y, float p2x, float p2y, float p3x, float p3y,
                                               float p4x, float p4y) {
  float det = (p4y - p3y) * (p2x - p1x) - (p4x - p3x) * (p2y - p1y);
  // (nearly) parallel segments never cross; the relative test keeps the decision out of fp32 rounding noise
  // at wrap onset, where both tangent segments lie along the chord
  float n12 = (p2x - p1x) * (p2x - p1x) + (p2y - p1y) * (p2y - p1y), n34 = (p4x - p3x) * (p4x - p3x) + (p4y - p3y) * (p4y - p3y);
  if (fabsf(det) < MINVALF || det * det < 4e-6f * n12 * n34) return false;
  float a = ((p4x - p3x) * (p1y - p3y) - (p4y - p3y) * (p1x - p3x)) / det;
  float b = ((p2x - p1x) * (p1y - p3y) - (p2y - p1y) * (p1x - p3x)) / det;
  return a >= 0.f && a <= 1.f && b >= 0.f && b <= 1.f;
}

__device__ __forceinline__ float wrap_circle(float pnt[4], float d0x, float d0y, float d1x, float d1y, bool has_side,
                                             float sdx, float sdy, float radius) {
  float sqlen0 = d0x * d0x + d0y * d0y, sqlen1 = d1x * d1x + d1y * d1y, sqrad = radius * radius;
  float difx = d1x - d0x, dify = d1y - d0y;
  float dd = difx * difx + dify * dify;
  float aa = clampf(-(difx * d0x + dify * d0y) / fmaxf(dd, MINVALF), 0.f, 1.f);
  float tx = d0x + aa * difx, ty = d0y + aa * dify;
  if (tx * tx + ty * ty > sqrad && (!has_side || sdx * tx + sdy * ty >= 0.f)) return -1.f;
  if (sqlen0 < sqrad || sqlen1 < sqrad) return -1.f;
  float sqrt0 = sqrtf(sqlen0 - sqrad), sqrt1 = sqrtf(sqlen1 - sqrad);
  float s0[4], s1[4], good0, good1;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float sgn = i == 0 ? 1.f : -1.f;
    float* sol = i == 0 ? s0 : s1;
    sol[0] = (d0x * sqrad + sgn * radius * d0y * sqrt0) / sqlen0;
    sol[1] = (d0y * sqrad - sgn * radius * d0x * sqrt0) / sqlen0;
    sol[2] = (d1x * sqrad - sgn * radius * d1y * sqrt1) / sqlen1;
    sol[3] = (d1y * sqrad + sgn * radius * d1x * sqrt1) / sqlen1;
    float good;
    if (has_side) {
      float ux = sol[0] + sol[2], uy = sol[1] + sol[3];
      float n = fmaxf(sqrtf(ux * ux + uy * uy), MINVALF);
      good = (ux * sdx + uy * sdy) / n;
    } else {
      float ux = sol[0] - sol[2], uy = sol[1] - sol[3];
      good = -(ux * ux + uy * uy);
    }
    if (seg_intersect(d0x, d0y, sol[0], sol[1], d1x, d1y, sol[2], sol[3])) good = -10000.f;
    if (i == 0) good0 = good; else good1 = good;
  }
  bool pick0 = good0 > good1;
#pragma unroll
  for (int k = 0; k < 4; k++) pnt[k] = pick0 ? s0[k] : s1[k];
  if (seg_intersect(d0x, d0y, pnt[0], pnt[1], d1x, d1y, pnt[2], pnt[3])) return -1.f;
  float c = clampf((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqrad, -1.f, 1.f);
  return radius * acosf(c);
}

__device__ __forceinline__ float wrap_geom(V3& w0, V3& w1, V3 x0, V3 x1, V3 gpos, const M3& gmat, float radius,
                                           bool is_cyl, bool has_side, V3 side) {
  V3 p0 = mtv(gmat, x0 - gpos), p1 = mtv(gmat, x1 - gpos);
  float n0 = sqrtf(dot(p0, p0)), n1 = sqrtf(dot(p1, p1));
  if (n0 < MINVALF || n1 < MINVALF) return -1.f;
  V3 ax0, ax1;
  if (is_cyl) {
    ax0 = v3(1.f, 0.f, 0.f); ax1 = v3(0.f, 1.f, 0.f);
  } else {
    ax0 = (1.f / n0) * p0;
    V3 nrm = cross(p0, p1);
    float nn = sqrtf(dot(nrm, nrm));
    if (nn < MINVALF) {
      V3 e = v3(1.f, 0.f, 0.f);
      float m = fabsf(ax0.x);
      if (fabsf(ax0.y) < m) { e = v3(0.f, 1.f, 0.f); m = fabsf(ax0.y); }
      if (fabsf(ax0.z) < m) { e = v3(0.f, 0.f, 1.f); }
      nrm = cross(ax0, e);
      nn = sqrtf(dot(nrm, nrm));
    }
    nrm = (1.f / fmaxf(nn, MINVALF)) * nrm;
    ax1 = cross(nrm, ax0);
    ax1 = (1.f / fmaxf(sqrtf(dot(ax1, ax1)), MINVALF)) * ax1;
  }
  float d0x = dot(p0, ax0), d0y = dot(p0, ax1), d1x = dot(p1, ax0), d1y = dot(p1, ax1);
  float sdx = 0.f, sdy = 0.f;
  if (has_side) {
    V3 s = mtv(gmat, side - gpos);
    sdx = dot(s, ax0); sdy = dot(s, ax1);
    float n = fmaxf(sqrtf(sdx * sdx + sdy * sdy), MINVALF);
    sdx /= n; sdy /= n;
  }
  float pnt[4];
  float wlen = wrap_circle(pnt, d0x, d0y, d1x, d1y, has_side, sdx, sdy, radius);
  if (wlen < 0.f) return -1.f;
  V3 r0 = pnt[0] * ax0 + pnt[1] * ax1, r1 = pnt[2] * ax0 + pnt[3] * ax1;
  if (is_cyl) {
    float L0 = sqrtf((p0.x - pnt[0]) * (p0.x - pnt[0]) + (p0.y - pnt[1]) * (p0.y - pnt[1]));
    float L1 = sqrtf((p1.x - pnt[2]) * (p1.x - pnt[2]) + (p1.y - pnt[3]) * (p1.y - pnt[3]));
    float tot = fmaxf(L0 + wlen + L1, MINVALF);
    r0.z = p0.z + (p1.z - p0.z) * L0 / tot;
    r1.z = p0.z + (p1.z - p0.z) * (L0 + wlen) / tot;
    float h = fabsf(r1.z - r0.z);
    wlen = sqrtf(wlen * wlen + h * h);
  }
  w0 = mv(gmat, r0) + gpos;
  w1 = mv(gmat, r1) + gpos;
  return wlen;
}


// ---- capsule axis vs convex primitive (mmo_collision.inc: sd_box / sd_cylinder / sd_ellipsoid / seg_shape) -----------
__device__ __forceinline__ float sd_box(V3 s, V3 q, V3& grad) {
  V3 d = v3(fabsf(q.x) - s.x, fabsf(q.y) - s.y, fabsf(q.z) - s.z);
  const V3 sg = v3(q.x < 0.f ? -1.f : 1.f, q.y < 0.f ? -1.f : 1.f, q.z < 0.f ? -1.f : 1.f);
  if (d.x > 0.f || d.y > 0.f || d.z > 0.f) {
    V3 e = v3(fmaxf(d.x, 0.f), fmaxf(d.y, 0.f), fmaxf(d.z, 0.f));
    const float n2 = dot(e, e), in_ = __frsqrt_rn(n2);
    grad = v3(e.x * in_ * sg.x, e.y * in_ * sg.y, e.z * in_ * sg.z);
    return n2 * in_;
  }
  if (d.x >= d.y && d.x >= d.z) { grad = v3(sg.x, 0.f, 0.f); return d.x; }
  if (d.y >= d.z) { grad = v3(0.f, sg.y, 0.f); return d.y; }
  grad = v3(0.f, 0.f, sg.z); return d.z;
}
__device__ __forceinline__ float sd_cylinder(V3 s, V3 q, V3& grad) {
  const float r2 = q.x * q.x + q.y * q.y;
  const float irho = r2 > MINVALF ? __frsqrt_rn(r2) : 0.f, rho = r2 * irho, dr = rho - s.x, dz = fabsf(q.z) - s.y;
  const float rx = r2 > MINVALF ? q.x * irho : 1.f, ry = q.y * irho, sz = q.z < 0.f ? -1.f : 1.f;
  if (dr > 0.f && dz > 0.f) { const float n2 = dr * dr + dz * dz, in_ = __frsqrt_rn(n2); grad = v3(dr * rx * in_, dr * ry * in_, dz * sz * in_); return n2 * in_; }
  if (dr > dz) { grad = v3(rx, ry, 0.f); return dr; }
  grad = v3(0.f, 0.f, sz); return dz;
}
// `tw` carries the Lagrange multiplier between calls: consecutive query points along the capsule axis are close, so a
// warm-started Newton needs few iterations (F is convex and decreasing: from the right of the root the first step lands
// left of it and the rest converge monotonically).  tw = NaN requests a cold start.
__device__ __forceinline__ float sd_ellipsoid(V3 s, V3 q0, V3& grad, float& tw) {
  const float sv[3] = {s.x, s.y, s.z}, qi[3] = {q0.x, q0.y, q0.z};
  float q[3], sq[3], s2[3], f0 = -1.f, amin = sv[0];
  int imin = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    q[i] = fabsf(qi[i]) < 1e-9f ? (qi[i] < 0.f ? -1e-9f : 1e-9f) : qi[i];
    s2[i] = sv[i] * sv[i]; sq[i] = sv[i] * q[i];
    const float r = q[i] * __builtin_amdgcn_rcpf(sv[i]);
    f0 += r * r;
    if (sv[i] < amin) { amin = sv[i]; imin = i; }
  }
  const float tlo = f0 >= 0.f ? 0.f : -amin * amin + amin * fabsf(q[imin]);
  const bool cold = !(tw == tw);
  float t = cold ? tlo : fmaxf(tw, tlo);
  const int iters = cold ? 9 : 4;
  for (int it = 0; it < iters; it++) {
    float F = -1.f, dF = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) { const float ri = __builtin_amdgcn_rcpf(t + s2[i]), w = sq[i] * ri; F += w * w; dF -= 2.f * w * w * ri; }
    if (dF > -MINVALF) break;
    t = fmaxf(t - F * __builtin_amdgcn_rcpf(dF), tlo);
  }
  tw = t;
  float g[3], n2 = 0.f, d2 = 0.f;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const float ri = __builtin_amdgcn_rcpf(t + s2[i]);
    const float x = s2[i] * q[i] * ri;
    g[i] = q[i] * ri; n2 += g[i] * g[i]; d2 += (q[i] - x) * (q[i] - x);
  }
  const float inv = __frsqrt_rn(n2);
  grad = v3(g[0] * inv, g[1] * inv, g[2] * inv);
  return f0 >= 0.f ? sqrtf(d2) : -sqrtf(d2);
}
__device__ __forceinline__ float sd_shape(int type, V3 s, V3 q, V3& grad, float& tw) {
  if (type == MM_GEOM_BOX) return sd_box(s, q, grad);
  if (type == MM_GEOM_CYLINDER) return sd_cylinder(s, q, grad);
  return sd_ellipsoid(s, q, grad, tw);
}
// minimiser of the convex g(t) = sd(a + t u) on [-h, h]: bisection on the sign of g'(t) = grad.u; flat stretches are
// bracketed with a +-tau tolerance and their midpoint is used (same rule as the oracle)
struct SegHit { float sd, t; V3 g; };
__device__ __forceinline__ SegHit seg_shape_call(int type, V3 s, V3 a0, V3 u, float h);
__device__ __forceinline__ float seg_dg(int type, V3 s, V3 a0, V3 u, float t, float& tw) {
  V3 g;
  sd_shape(type, s, a0 + t * u, g, tw);
  return dot(g, u);
}
// POLISH: after the bisection, bracketed false-position steps on g' (monotone: g is convex) using the values at the ends of the
// bracket.  13 bisection steps leave t within h * 2^-13 ~ 4e-6 m of the root; the oracle bisects 40 times in fp64, and a contact
// point that sits 4e-6 m off moves the pyramid rows by ~1e-4 relative (round 2: constrained acceleration of the reorient batch
// only 3e-3 from the oracle).  Two secant steps inside the bracket take a smooth g' (ellipsoid, rounded edges) to fp32
// resolution; at a kink of g' (box / cylinder edge) they stay inside the bracket, so they are never worse than the bisection.
__device__ __forceinline__ float seg_bisect(int type, V3 s, V3 a0, V3 u, float lo, float hi, float thr, int iters, float& tw, int polish = 0) {
  float dlo = seg_dg(type, s, a0, u, lo, tw) - thr;
  if (dlo > 0.f) return lo;
  float dhi = seg_dg(type, s, a0, u, hi, tw) - thr;
  if (dhi <= 0.f) return hi;
  for (int it = 0; it < iters; it++) {
    const float mid = 0.5f * (lo + hi);
    const float dm = seg_dg(type, s, a0, u, mid, tw) - thr;
    if (dm > 0.f) { hi = mid; dhi = dm; } else { lo = mid; dlo = dm; }
  }
  float t = 0.5f * (lo + hi);
  for (int it = 0; it < polish; it++) {
    const float den = dhi - dlo;
    if (!(den > 1e-12f)) break;
    float tn = lo - dlo * (hi - lo) / den;
    tn = fminf(fmaxf(tn, lo), hi);
    const float dn = seg_dg(type, s, a0, u, tn, tw) - thr;
    t = tn;
    if (dn > 0.f) { hi = tn; dhi = dn; } else { lo = tn; dlo = dn; }
  }
  return t;
}
// same rule as the oracle's seg_shape (mmo_collision.inc): root of g', flat minima of box / cylinder replaced by the
// midpoint of their +-tau interval; 12 bisection steps + 3 bracketed secant steps on the root of g'
__device__ __forceinline__ float seg_shape(int type, V3 s, V3 a0, V3 u, float h, float& tbest, V3& grad) {
  const float tau = 1e-4f;
  float tw = __builtin_nanf("");
  float t = seg_bisect(type, s, a0, u, -h, h, 0.f, 12, tw, 3);
  if (type != MM_GEOM_ELLIPSOID) {
    const float dl = 0.02f * h;
    float tl = t, tr = t;
    if (seg_dg(type, s, a0, u, fmaxf(t - dl, -h), tw) > -tau) tl = seg_bisect(type, s, a0, u, -h, t, -tau, 13, tw);
    if (seg_dg(type, s, a0, u, fminf(t + dl, h), tw) <= tau) tr = seg_bisect(type, s, a0, u, t, h, tau, 13, tw);
    t = 0.5f * (tl + tr);
  }
  tbest = t;
  return sd_shape(type, s, a0 + t * u, grad, tw);
}

__device__ __forceinline__ SegHit seg_shape_call(int type, V3 s, V3 a0, V3 u, float h) {
  SegHit r;
  r.sd = seg_shape(type, s, a0, u, h, r.t, r.g);
  return r;
}

// ------------------------------------------------------------------ muscle model (A6)
__device__ __forceinline__ float muscle_fl(float L, float lmin, float lmax) {
  if (L < lmin || L > lmax) return 0.f;
  float a = 0.5f * (lmin + 1.f), b = 0.5f * (1.f + lmax), x;
  if (L <= a) { x = (L - lmin) / fmaxf(MINVALF, a - lmin); return 0.5f * x * x; }
  if (L <= 1.f) { x = (1.f - L) / fmaxf(MINVALF, 1.f - a); return 1.f - 0.5f * x * x; }
  if (L <= b) { x = (L - 1.f) / fmaxf(MINVALF, b - 1.f); return 1.f - 0.5f * x * x; }
  x = (lmax - L) / fmaxf(MINVALF, lmax - b);
  return 0.5f * x * x;
}
__device__ __forceinline__ float muscle_f0(const float* prm, float acc0) {
  return prm[2] >= 0.f ? prm[2] : prm[3] / fmaxf(MINVALF, acc0);
}
__device__ __forceinline__ float muscle_gain(float len, float vel, float lr0, float lr1, float acc0, const float* prm) {
  float force = muscle_f0(prm, acc0);
  float L0 = (lr1 - lr0) / fmaxf(MINVALF, prm[1] - prm[0]);
  float L = prm[0] + (len - lr0) / fmaxf(MINVALF, L0);
  float V = vel / fmaxf(MINVALF, L0 * prm[6]);
  float FL = muscle_fl(L, prm[4], prm[5]);
  float fvmax = prm[8], y = fvmax - 1.f, FV;
  if (V <= -1.f) FV = 0.f;
  else if (V <= 0.f) FV = (V + 1.f) * (V + 1.f);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / fmaxf(MINVALF, y);
  else FV = fvmax;
  return -force * FL * FV;
}
__device__ __forceinline__ float muscle_bias(float len, float lr0, float lr1, float acc0, const float* prm) {
  float force = muscle_f0(prm, acc0);
  float L0 = (lr1 - lr0) / fmaxf(MINVALF, prm[1] - prm[0]);
  float L = prm[0] + (len - lr0) / fmaxf(MINVALF, L0);
  float b = 0.5f * (1.f + prm[5]), fpmax = prm[7], x;
  if (L <= 1.f) return 0.f;
  if (L <= b) { x = (L - 1.f) / fmaxf(MINVALF, b - 1.f); return -force * fpmax * 0.5f * x * x; }
  x = (L - b) / fmaxf(MINVALF, b - 1.f);
  return -force * fpmax * (0.5f + x);
}
__device__ __forceinline__ float sigmoid5(float x) {
  if (x <= 0.f) return 0.f;
  if (x >= 1.f) return 1.f;
  return x * x * x * (3.f * x * (2.f * x - 5.f) + 10.f);
}
__device__ __forceinline__ float muscle_dynamics(float ctrl, float act, const float* prm) {
  float cc = clampf(ctrl, 0.f, 1.f), ac = clampf(act, 0.f, 1.f);
  float tau_act = prm[0] * (0.5f + 1.5f * ac), tau_deact = prm[1] / (0.5f + 1.5f * ac);
  float dctrl = cc - act, tau;
  if (prm[2] < MINVALF) tau = dctrl > 0.f ? tau_act : tau_deact;
  else tau = tau_deact + (tau_act - tau_deact) * sigmoid5(dctrl / prm[2] + 0.5f);
  return dctrl / fmaxf(MINVALF, tau);
}

#ifndef MM_STAGE_PROF
#define MM_STAGE_PROF 0   /* 1: in-kernel stage timers (mm_debug_set_prof); a tools build (tools/build_variant.py prof -DMM_STAGE_PROF=1): \
                             the 26 SGPRs of the timer array and the clock reads cost the product kernels 1-2 % */
#endif
#ifndef MM_SPARSE_GEN
#define MM_SPARSE_GEN 0   /* 1: the general-row kernels use the tree-sparse solve for the two M solves of a pass (solve0, Euler).  Measured: \
                             the extra live state tips these 256-VGPR kernels into 80 spills; reorient -3 %, self-contact hand -5 % */
#endif
#ifndef MM_FOLD_RESET
#define MM_FOLD_RESET 1   /* 0: no folded walk / reorient reset (A/B switch; MM_INFO_FOLDED_RESET then reports 0) */
#endif
#ifndef MM_LS_RELSTOP
/* Experiment, OFF: end the exact line search once a turn moves alpha by less than MM_LS_RELSTOP_TOL relative, instead of running the
   safeguarded Newton on phi'(alpha) until its step vanishes in float resolution.  The stage timers put ~10 k of the hand's 32 k
   Newton cycles per pass in the search, but the precision of alpha is not slack: qacc += alpha * search with |search| up to
   1e2...1e3, and tolerance 1e-5 took the north-star count from 61 to 55 of 64 envs (median 3.6e-6 -> 1.2e-5) for +1.9 %
   throughput; 1e-6: 58 of 64, +0.8 %; 3e-7: 61 of 64, +0.6 % (profiles/r03_north_star_ab.json).  Not worth a digit. */
#define MM_LS_RELSTOP 0
#endif
#ifndef MM_LS_RELSTOP_TOL
#define MM_LS_RELSTOP_TOL 3e-7f
#endif
#ifndef MM_NEWTON_POLISH
#define MM_NEWTON_POLISH 0   /* experiment (limit-rows-only kernels): one extra Newton step after the convergence test fires */
#endif
#ifndef MM_NEWTON_TRUE_MV
#define MM_NEWTON_TRUE_MV 0  /* experiment (tree-sparse kernels): M search as an explicit product instead of -grad - D search */
#endif
#ifndef MM_SPARSE_LDL
#define MM_SPARSE_LDL 1   /* 0: dense register Cholesky in every kernel (A/B switch) */
#endif
// =========================================================================== engine
/* keep a wave-uniform value in an SGPR: opaque to rematerialisation (an s_load + s_waitcnt at every use).  The readfirstlane
   folds away when the value already sits in an SGPR; without it the backend dies with "illegal VGPR to SGPR copy" in the
   instantiations where it had moved the (uniform) value's computation to the vector ALU. */
#define PIN_S(x) do { (x) = __builtin_amdgcn_readfirstlane(x); asm volatile("" : "+s"(x)); } while (0)
#define AI_(o) (Tab<int>{mb, (uint32_t)(o)})
#define AF_(o) (Tab<float>{mb, (uint32_t)(o)})
// What a lane knows about the dof-tree segment it owns (sp_factor_solve): depth range [t, b] of the segment, its step in the
// elimination order (-1: the lane owns none), its index (slot of its update matrix), the dof ids on the path root .. bottom by
// depth, the child segment indices (0xff = none); depth = depth of the lane's own dof (-1: no dof).
struct SegLane {
  int t, b, lv, id, depth;
  unsigned path_lo, path_hi, ch_lo, ch_hi;
};
// All member functions are collective over the G lanes of one env group.  NVP = padded nv (compile time).
// INTEG: 0 semi-implicit Euler (eulerdamp), 1 RK4, 2 implicitfast (compile-time variants: each one's state machine would cost
// the others registers)
template <int G, int NVP, bool GEN, int INTEG>
struct Engine {
  static constexpr bool RK4 = INTEG == 1, IMPL = INTEG == 2;
  // Two waves per env group (a.two_wave; the Euler kernels): when the batch leaves SIMDs empty -- leg-walk at 1024 envs is one
  // wave per SIMD, the elbow at 4096 envs half a wave, all of them waiting on dependent latency most of the time -- a second
  // wave of the block (same lanes, same envs) runs the stages that need no per-lane register state (tendon paths + Jacobian,
  // tendon velocities, muscle / actuator forces, J'f) concurrently with the main wave's constraint assembly, velocity / RNE
  // stage, CRB and factorisation, and -- dense kernels -- factorises M + h B for the Euler step while the main wave is in
  // Newton.  The two meet through LDS counters per env with bounded spin waits (a lost partner raises status bit 16
  // instead of hanging).
  static constexpr bool TW = INTEG != 1;   // Euler and implicitfast (RK4's four forward passes per step would need the helper's pass logic)
  static constexpr int TW_DONE = 0x7fffffff;
  int tw_n;     // forward passes opened so far (two-wave launches)
  int o_tile;   // LDS word offset of the dense tile factor_core / solve work on (u1; the helper wave's own tile in two-wave launches)
  // Dense Cholesky form.  Left-looking (row j of L from an LDS tile, one pivot broadcast per column) executes ~40 % fewer
  // instructions than right-looking (NVP^2 / 2 cross-lane broadcasts) but adds an LDS round trip per column.  Groups narrower than
  // the wave always take it (a broadcast costs ~5 issue slots there).  One env per wave: it wins where two or more waves per SIMD
  // keep the issue ports busy (hand-family models at their batch sizes: reorient +3.4 %) and loses where a lone wave per SIMD
  // waits on latency (leg-walk at 1024 envs: -7 %); the tile width stands in for that distinction.
#ifndef MM_LEFT_LOOKING_MAX
#define MM_LEFT_LOOKING_MAX 32   /* widest one-env-per-wave tile factorised left-looking (A/B: 36 = the leg too) */
#endif
  static constexpr bool LEFT_LOOKING = NVP >= 8 && (G < 64 || NVP <= MM_LEFT_LOOKING_MAX);
  // M x and J x with x through an LDS vector (one write, NVP / 4 broadcast 128-bit reads) instead of NVP cross-lane broadcasts:
  // always for narrow groups; one env per wave, measured per tile width: 24-wide (self-contact hand) +1.2 %, 32-wide (reorient) -2.7 %
  static constexpr bool LDS_VECTOR = NVP >= 8 && (G < 64 || NVP <= 24);
  const KArgs& a;
  const KConst& kc;    // model constants of the launch (see MM_CONST_IN_REGS)
  const uint32_t* mb;  // model words (LDS-resident copy or global)
  unsigned long long pf[MM_STAGE_PROF ? NPROF : 1];
  float* W;     // LDS tables of this env
  const int g;  // lane within group == owned body / dof index
  int status;   // sticky status bits (group-uniform)
  int nefc, niter;
  // ---- body-lane registers (valid for g < nbody)
  V3 b_xpos, b_xipos;
  Q4 b_xquat;
  float b_cinert[10];
  float b_cvel[6];
  int b_depth, b_parent;
  // integer model constants of the owned body and of its first two joints (loaded once per kernel); the float constants
  // (body / joint frames) are read from the LDS-resident model where they are used: holding them cost 23 VGPRs and spills
  int c_jn, c_ja;
  int c_rowj;         // joint of dof g (its limit row lives in lane c_rowj); c_rowj_mine: dof g is that joint's (first) dof
  bool c_rowj_mine;
  // ---- dof-lane registers (valid for g < nv)
  float d_cdof[6];
  float d_qvel, d_warm, d_bias, d_smooth, d_qaccsm, d_qacc, d_qfrccon;
  float Mrow[NVP];   // row g of M (dense, symmetric)
  float Lrow[NVP];   // row g of the current Cholesky factor  (L[g][k], k <= g)
  float d_dinv;      // 1 / L[g][g]   (SP: 1 / D[g])
  // Tree-sparse storage (SP kernels: limit rows only, so every matrix that gets factorised -- M, M + h B, M + diag(D_active) --
  // has M's pattern: non-zero only between a dof and its ancestors).  Lane g keeps its row indexed by the ABSOLUTE depth of the
  // ancestor: Ms[e] = M[g][ancestor of g at depth e] for e < d_depth (0 beyond), Md = M[g][g].  (A descendant's row and its
  // ancestor's row then agree on the index of every common ancestor, so an ancestor reads a descendant's row with aligned
  // 128-bit loads.)
  static constexpr int SD = 8;     // maximum depth of the dof tree of an SP model (host routes deeper ones to the GEN kernels)
  static constexpr int TS = 12;    // row stride of the published rows: [row 0..7, 1/D, rhs, -, -]
  static constexpr bool SP = MM_SPARSE_LDL && !GEN && NVP >= 8 && INTEG != 2;
  // Per-body records that live in the u1 scratch get ODD strides: lane g touches record g, and a ds_read_b32 of 32 lanes hits 32
  // banks -- stride 4 (quaternions) and 12 (cvel | cacc) were 4-way bank conflicts on every access of the pointer-jumping rounds
  // (8 of them per forward pass).  u1 has the room (host: >= (CVS + 1) * nbody words), so this costs no LDS.
#ifndef MM_QS
#define MM_QS 5
#define MM_CVS 13
#endif
  static constexpr int QS = MM_QS, CVS = MM_CVS;
  // Row stride of the dense NVP x NVP LDS tile(s).  Lane i works on ROW i, so a row stride that is a multiple of 32 words puts the
  // lanes of a wave on one bank: with NVP = 32 the left-looking factor's column store T[i][j] was a 32-way conflict and the row
  // read-backs (M after CRB, J'DJ after the MFMA product) 8-way; rocprofv3 had 47 % of the reorient kernel's LDS cycles as
  // bank-conflict cycles (profiles/r03a_pmc.json).  Stride 36 (stride / 4 odd): a 128-bit row access of 16 lanes covers all 64
  // banks once, the column store is 4-way: reorient kernel 0.761 -> 0.719 ms.  Only the 32-wide tile is padded: for the 24- and
  // 36-wide ones (8-way / 4-way conflicts) the extra LDS words cost the self-contact hand its LDS-resident model copy and the
  // leg its two-wave launch (measured: -4 % / -3 %), which outweighs the conflicts.
  static constexpr int TD = (!SP && NVP == 32) ? 36 : NVP;
  float Ms[SD], Md;
  unsigned anc_lo, anc_hi;   // ancestor dof ids by absolute depth, one byte each: depth 0..3 | 4..6
  int d_depth;               // depth of dof g in the dof tree (0 = no parent dof); -1 on lanes without a dof
  SegLane sgl;
  // ---- joint-limit row owned by this lane (lower side: lanes < G/2, upper side: lanes >= G/2)
  bool r_active;
  float r_D, r_aref, r_sign, r_jar;
  int r_dof;
  // ---- general rows (GEN): lane r owns row r of efc_J (LDS); equality rows are always active
  bool r_eq;
  float r_floss;    // friction-loss row: bound of the row force (0 on every other row)
  int nrows_wave;   // wave-uniform upper bound of nefc over the envs of this wave
  float env_gsv[3];         // this env's row of mm_state.geom_size_env (values: the folded reorient reset rewrites them)
  bool env_has_gs;
  float rk_v0, rk_vsum, rk_asum;   // RK4: qvel at the start of the step, weighted sums of stage qvel / qacc
  int env_gtype;            // this env's entry of mm_state.geom_type_env (or -1)
  int env;                  // env index (per-env model deltas on a body: mm_state.body_mass_env / body_pos_env)

  __device__ __forceinline__ Engine(const KArgs& a_, const KConst& kc_, const uint32_t* mb_, float* W_, int g_)
      : a(a_), kc(kc_), mb(mb_), W(W_), g(g_), status(0), nefc(0), niter(0), tw_n(0), o_tile(0) {
    o_tile = KL().u1;   // (not in the initialiser list: the member is declared ahead of the references KL() goes through)
#pragma unroll
    for (int i = 0; i < (MM_STAGE_PROF ? NPROF : 1); i++) pf[i] = 0;
    d_warm = 0.f; d_qvel = 0.f;
    b_depth = (g < KD().nbody) ? AUXI(body_depth)[g] : -1;
    b_parent = (g > 0 && g < KD().nbody) ? MI_(BODY_PARENT)[g] : 0;
    {
      const bool isb = g > 0 && g < KD().nbody;
      c_ja = isb ? MI_(BODY_JNTADR)[g] : 0;
      c_jn = isb ? MI_(BODY_JNTNUM)[g] : 0;
      c_rowj = g < KD().nv ? MI_(DOF_JNTID)[g] : 0;
      c_rowj_mine = g < KD().nv && MI_(JNT_DOFADR)[c_rowj] == g;
    }
    r_dof = 0; r_active = false; r_sign = 1.f; r_D = 0.f; r_aref = 0.f; r_jar = 0.f; r_eq = false; r_floss = 0.f; nrows_wave = 0; env_has_gs = false; env_gsv[0] = env_gsv[1] = env_gsv[2] = 0.f; env_gtype = -1; rk_v0 = rk_vsum = rk_asum = 0.f;
    // lanes that own no body / dof still take part in reductions with zero weights: their registers must
    // hold finite values (0 * garbage could be NaN)
    b_xpos = v3(0.f, 0.f, 0.f); b_xipos = b_xpos;
    Q4 qi = {1.f, 0.f, 0.f, 0.f};
    b_xquat = qi;
#pragma unroll
    for (int k = 0; k < 10; k++) b_cinert[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) { b_cvel[k] = 0.f; d_cdof[k] = 0.f; }
    d_bias = d_smooth = d_qaccsm = d_qacc = d_qfrccon = 0.f; d_dinv = 1.f;
#pragma unroll
    for (int k = 0; k < NVP; k++) { Mrow[k] = 0.f; Lrow[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < SD; k++) Ms[k] = 0.f;
    Md = 1.f;
    anc_lo = anc_hi = 0u; d_depth = -1;
    sgl = seg_lane_none();
    if constexpr (SP) {
      if (g < KD().nv) {
        const int* dpar = MI_(DOF_PARENTID);
        // (fixed trip counts: a data-dependent loop over the model table here runs into a backend error, "illegal VGPR to
        // SGPR copy", in the LDS-model variants)
        int dep = 0, j = dpar[g];
#pragma unroll
        for (int k = 0; k < SD - 1; k++)
          if (j >= 0) { dep++; j = dpar[j]; }
        d_depth = dep;
        j = dpar[g];
#pragma unroll
        for (int k = 0; k < SD - 1; k++)
          if (j >= 0) {
            const int e = dep - 1 - k;
            if (e < 4) anc_lo |= (unsigned)j << (8 * e); else anc_hi |= (unsigned)j << (8 * (e - 4));
            j = dpar[j];
          }
        sgl = seg_lane_load();
      }
    }
  }
  // Back to the constructor's values for everything a forward pass recomputes (the folded reset calls this before its second
  // forward pass, so that none of these registers is live across the task stage and the reset block)
  __device__ __forceinline__ void reinit_transients() {
    r_dof = 0; r_active = false; r_sign = 1.f; r_D = 0.f; r_aref = 0.f; r_jar = 0.f; r_eq = false; r_floss = 0.f; nrows_wave = 0; rk_v0 = rk_vsum = rk_asum = 0.f;
    b_xpos = v3(0.f, 0.f, 0.f); b_xipos = b_xpos;
    Q4 qi = {1.f, 0.f, 0.f, 0.f};
    b_xquat = qi;
#pragma unroll
    for (int k = 0; k < 10; k++) b_cinert[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) { b_cvel[k] = 0.f; d_cdof[k] = 0.f; }
    d_bias = d_smooth = d_qaccsm = d_qacc = d_qfrccon = 0.f; d_dinv = 1.f;
#pragma unroll
    for (int k = 0; k < NVP; k++) { Mrow[k] = 0.f; Lrow[k] = 0.f; }
#pragma unroll
    for (int k = 0; k < SD; k++) Ms[k] = 0.f;
    Md = 1.f;
    nefc = 0; niter = 0;
  }
  __device__ __forceinline__ static SegLane seg_lane_none() {
    SegLane q; q.t = q.b = 0; q.lv = -1; q.id = 0; q.depth = -1; q.path_lo = q.path_hi = 0u; q.ch_lo = q.ch_hi = 0xffffffffu;
    return q;
  }
  // this lane's row of Aux.dof_seg (lanes without a dof own nothing)
  __device__ __forceinline__ SegLane seg_lane_load() const {
    SegLane q = seg_lane_none();
    if (g < KD().nv) {
      const int* sg = AUXI(dof_seg) + 6 * g;
      const int w = sg[0];
      q.depth = sg[5];
      if (w >= 0) {
        q.t = w & 15; q.b = (w >> 4) & 15; q.lv = (w >> 8) & 15; q.id = (w >> 16) & 255;
        q.path_lo = (unsigned)sg[1]; q.path_hi = (unsigned)sg[2]; q.ch_lo = (unsigned)sg[3]; q.ch_hi = (unsigned)sg[4];
      }
    }
    return q;
  }
  // dof id of this lane's ancestor at absolute depth E_ (valid for E_ < d_depth)
  template <int E_>
  __device__ __forceinline__ int anc() const {
    static_assert(E_ >= 0 && E_ < SD - 1, "ancestor depth");
    return (int)(E_ < 4 ? (anc_lo >> (8 * E_)) & 255u : (anc_hi >> (8 * (E_ - 4))) & 255u);
  }

  __device__ __forceinline__ V3 origin() const { return v3(KD().ox, KD().oy, KD().oz); }
  __device__ __forceinline__ float com_of_body(int b, int k) const { return W[KL().com + 3 * AUXI(body_rootslot)[b] + k]; }

  __device__ __forceinline__ void tw_signal(int idx, int n) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (g == 0) reinterpret_cast<volatile int*>(W + KL().flags)[idx] = n;
  }
  __device__ __forceinline__ int tw_wait(int idx, int n) {
    volatile int* f = reinterpret_cast<volatile int*>(W + KL().flags);
    int v = f[idx];
    for (int it = 0; v < n && it < (1 << 22); it++) { __builtin_amdgcn_s_sleep(1); v = f[idx]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (v < n) status |= 16;   // the partner wave never arrived
    return v;
  }
  // the helper wave of a two-wave launch: for every forward pass the main wave opens, the tendon and actuation stages
  __device__ __forceinline__ void helper_loop() {
    for (int n = 1;; n++) {
      const int v = tw_wait(0, n);
      if (v == TW_DONE || v < n) break;
      tendon();
      tendon_velocity();
      actuation();
      tw_signal(1, n);
      if constexpr (IMPL) { implicit_w(KL().mtile); tw_signal(3, n); }
      if (!SP && !IMPL && KD().any_damping && KD().eulerdamp) {
        // the factor of M + h B the Euler step will need (mj_Euler's implicit joint damping) does not depend on the constraint
        // solve: computed here while the main wave is in Newton.  M arrives in the second tile and L leaves in it.
        tw_wait(2, n);
        const float* Mg = W + KL().mtile + (g < NVP ? g : 0) * TD;
#pragma unroll
        for (int k4 = 0; k4 < NVP / 4; k4++) {
          const float4 r = *reinterpret_cast<const float4*>(Mg + 4 * k4);
          Mrow[4 * k4] = g < NVP ? r.x : 0.f; Mrow[4 * k4 + 1] = g < NVP ? r.y : 0.f;
          Mrow[4 * k4 + 2] = g < NVP ? r.z : 0.f; Mrow[4 * k4 + 3] = g < NVP ? r.w : 0.f;
        }
        GSYNC();
        o_tile = KL().mtile;
        factor(g < KD().nv ? KD().timestep * MF_(DOF_DAMPING)[g] : 0.f);
        if (g < NVP) W[KL().mtile + NVP * TD + g] = d_dinv;   // 1 / L[g][g] as computed (not re-derived from L: bit-identical solves)
        tw_signal(3, n);
      }
    }
  }

  // ---------------------------------------------------------------- A1 kinematics
  __device__ __forceinline__ void kinematics() {
    // offsets read once and pinned in SGPRs for this stage (see PIN_S)
    int o_xpos = KL().xpos; PIN_S(o_xpos); int o_u1 = KL().u1; PIN_S(o_u1); int o_xmat = KL().xmat; PIN_S(o_xmat); int o_xanchor = KL().xanchor; PIN_S(o_xanchor); int o_xaxis = KL().xaxis; PIN_S(o_xaxis); int o_qpos = KL().qpos; PIN_S(o_qpos); int s_JNT_POS = SECOFF_(JNT_POS); PIN_S(s_JNT_POS); int s_JNT_AXIS = SECOFF_(JNT_AXIS); PIN_S(s_JNT_AXIS); int s_QPOS0 = SECOFF_(QPOS0); PIN_S(s_QPOS0); int s_BODY_POS = SECOFF_(BODY_POS); PIN_S(s_BODY_POS); int s_BODY_QUAT = SECOFF_(BODY_QUAT); PIN_S(s_BODY_QUAT); int s_BODY_IPOS = SECOFF_(BODY_IPOS); PIN_S(s_BODY_IPOS); int d_nlevel_ = KD().nlevel; PIN_S(d_nlevel_); int d_nbody_ = KD().nbody; PIN_S(d_nbody_);
    const auto& L = KL();
    const V3 org = origin();
    if (g == 0) {
      b_xpos = -1.f * org; b_xipos = b_xpos;   // the world body (and everything attached to it) in the internal frame
      Q4 q = {1.f, 0.f, 0.f, 0.f};
      b_xquat = q;
      st3(W + o_xpos, b_xpos);
      W[o_u1] = 1.f; W[o_u1 + 1] = 0.f; W[o_u1 + 2] = 0.f; W[o_u1 + 3] = 0.f;
      for (int k = 0; k < 9; k++) W[o_xmat + k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
    }
    GSYNC();
    // Phase A (all bodies at once): transform of every body in its PARENT's frame, its joints applied; joint anchors / axes
    // are left in LDS in that frame.  Phase B: pointer jumping -- every body composes its transform with the one of the
    // ancestor it currently points at and then points at that ancestor's target: ceil(log2(depth)) rounds instead of one
    // round per tree level (4 instead of 9 for the hand).  Phase C: world anchors / axes from the parent's final frame.
    const int nb = d_nbody_;
    const bool isb = g > 0 && g < nb;
    V3 tp = g == 0 ? -1.f * org : v3(0.f, 0.f, 0.f);   // lane 0 republishes the world body's frame in every round below
    Q4 tq = {1.f, 0.f, 0.f, 0.f};
    if (isb) {
      const int b = g;
      V3 pos = (a.s.body_pos_env && b == a.s.body_pos_env_id) ? ld3(a.s.body_pos_env + (size_t)env * 3) : ld3(AF_(s_BODY_POS) + 3 * b);
      if (b_parent == 0) pos = pos - org;
      Q4 quat = ldq(AF_(s_BODY_QUAT) + 4 * b);
      // The constants of joint i + 1 are requested before joint i is worked on (a body with several joints is a serial chain in
      // one lane; with the model read through L2 every joint paid type -> qpos address -> qpos0 as dependent misses)
      const int* JP = AUXI(jnt_pack);
      const int j0_ = c_ja > 0 ? c_ja : 0;
      int pk_ = JP[2 * j0_]; float pq0_ = __int_as_float(JP[2 * j0_ + 1]);
      V3 pjpos_ = ld3(AF_(s_JNT_POS) + 3 * j0_), pjax_ = ld3(AF_(s_JNT_AXIS) + 3 * j0_);
      for (int i = 0; i < c_jn; i++) {
        const int j = c_ja + i;
        const int type = pk_ & 15, qa = (pk_ >> 14) & 1023;
        const V3 jpos = pjpos_, jax = pjax_;
        const float q0 = pq0_;
        {
          const int jn_ = i + 1 < c_jn ? j + 1 : j;
          pk_ = JP[2 * jn_]; pq0_ = __int_as_float(JP[2 * jn_ + 1]);
          pjpos_ = ld3(AF_(s_JNT_POS) + 3 * jn_); pjax_ = ld3(AF_(s_JNT_AXIS) + 3 * jn_);
        }
        if (type == MM_JNT_FREE) {      // child of the world: its frame is the world frame
          pos = ld3(W + o_qpos + qa) - org;
          quat = qnorm(ldq(W + o_qpos + qa + 3));
          st3(W + o_xanchor + 3 * j, pos);
          M3 m = q2m(quat);
          st3(W + o_xaxis + 3 * j, v3(m.m[2], m.m[5], m.m[8]));
          continue;
        }
        M3 m = q2m(quat);
        V3 anchor = pos + mv(m, jpos), axis = mv(m, jax);
        st3(W + o_xanchor + 3 * j, anchor);
        st3(W + o_xaxis + 3 * j, axis);
        if (type == MM_JNT_SLIDE) {
          pos = pos + (W[o_qpos + qa] - q0) * axis;
        } else if (type == MM_JNT_HINGE) {
          float ang = W[o_qpos + qa] - q0;
          float sn, cs;
          sincos_small(0.5f * ang, &sn, &cs);
          Q4 ql = {cs, jax.x * sn, jax.y * sn, jax.z * sn};
          quat = qmul(quat, ql);
          pos = anchor - mv(q2m(quat), jpos);
        } else {  // ball
          quat = qmul(quat, qnorm(ldq(W + o_qpos + qa)));
          pos = anchor - mv(q2m(quat), jpos);
        }
      }
      tp = pos; tq = qnorm(quat);
    }
    int up = isb ? b_parent : 0;
    int* UP = reinterpret_cast<int*>(W + o_xmat);     // scratch: xmat is written last
    int nround = 0;
    for (int s_ = 1; s_ < d_nlevel_; s_ <<= 1) nround++;
    for (int r = 0; r < nround; r++) {
      if (g < nb) {
        st3(W + o_xpos + 3 * g, tp);
        W[o_u1 + QS * g] = tq.w; W[o_u1 + QS * g + 1] = tq.x; W[o_u1 + QS * g + 2] = tq.y; W[o_u1 + QS * g + 3] = tq.z;
        UP[g] = up;
      }
      GSYNC();
      if (up > 0) {
        const V3 pp = ld3(W + o_xpos + 3 * up);
        const Q4 pq = ldq(W + o_u1 + QS * up);
        const int uu = UP[up];
        tp = pp + mv(q2m(pq), tp);
        tq = qmul(pq, tq);
        up = uu;
      }
      GSYNC();
    }
    // final frames
    if (isb) {
      tq = qnorm(tq);
      st3(W + o_xpos + 3 * g, tp);
      W[o_u1 + QS * g] = tq.w; W[o_u1 + QS * g + 1] = tq.x; W[o_u1 + QS * g + 2] = tq.y; W[o_u1 + QS * g + 3] = tq.z;
      b_xpos = tp; b_xquat = tq;
    }
    GSYNC();
    if (isb) {
      const M3 m = q2m(tq);
      b_xipos = tp + mv(m, ld3(AF_(s_BODY_IPOS) + 3 * g));
      // joint anchors / axes: parent frame -> world (a free joint's parent is the world: nothing to do)
      const int p = b_parent;
      if (p > 0) {
        const V3 pp = ld3(W + o_xpos + 3 * p);
        const M3 pm = q2m(ldq(W + o_u1 + QS * p));
        for (int i = 0; i < c_jn; i++) {
          const int j = c_ja + i;
          st3(W + o_xanchor + 3 * j, pp + mv(pm, ld3(W + o_xanchor + 3 * j)));
          st3(W + o_xaxis + 3 * j, mv(pm, ld3(W + o_xaxis + 3 * j)));
        }
      }
    }
    GSYNC();   // UP scratch (xmat region) is dead for everybody: write the rotation matrices
    if (g == 0)
      for (int k = 0; k < 9; k++) W[o_xmat + k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
    if (isb) {
      const M3 m = q2m(tq);
      for (int k = 0; k < 9; k++) W[o_xmat + 9 * g + k] = m.m[k];
    }
    GSYNC();
  }

  __device__ __forceinline__ V3 site_pos(int s) const {
    int b = MI_(SITE_BODYID)[s];
    return ld3(W + KL().xpos + 3 * b) + mv(ldm(W + KL().xmat + 9 * b), ld3(MF_(SITE_POS) + 3 * s));
  }
  __device__ __forceinline__ V3 geom_pos(int gi) const {
    int b = MI_(GEOM_BODYID)[gi];
    return ld3(W + KL().xpos + 3 * b) + mv(ldm(W + KL().xmat + 9 * b), ld3(MF_(GEOM_POS) + 3 * gi));
  }
  __device__ __forceinline__ M3 geom_mat(int gi) const {  // xmat_body * R(geom_quat)
    int b = MI_(GEOM_BODYID)[gi];
    M3 A = ldm(W + KL().xmat + 9 * b), B = q2m(ldq(MF_(GEOM_QUAT) + 4 * gi)), R;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) R.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return R;
  }

  // subtree COM of each tree root, body inertias about it (registers), dof motion axes (LDS + registers)
  __device__ __forceinline__ void com_pos() {
    const auto& L = KL();
    const int nb = KD().nbody;
    const bool isb = g > 0 && g < nb;
    float ms = isb ? MF_(BODY_MASS)[g] : 0.f;
    if (a.s.body_mass_env && g == a.s.body_mass_env_id) ms = a.s.body_mass_env[env];
    int myslot = isb ? AUXI(body_rootslot)[g] : -1;
    for (int r = 0; r < KX().nroot; r++) {
      float w = (myslot == r) ? ms : 0.f;
      float sm = gsum<G>(w), sx = gsum<G>(w * b_xipos.x), sy = gsum<G>(w * b_xipos.y), sz = gsum<G>(w * b_xipos.z);
      if (g == 0) {
        int rb = AUXI(root_list)[r];
        V3 c;
        if (sm < MINVALF) c = ld3(W + L.xpos + 3 * rb);
        else c = (1.f / sm) * v3(sx, sy, sz);
        st3(W + L.com + 3 * r, c);
      }
    }
    GSYNC();
    if (isb) {
      V3 c = ld3(W + L.com + 3 * myslot);
      M3 R = q2m(qmul(b_xquat, ldq(MF_(BODY_IQUAT) + 4 * g)));
      V3 I = ld3(MF_(BODY_INERTIA) + 3 * g);
      V3 r = b_xipos - c;
      float xx = 0.f, yy = 0.f, zz = 0.f, xy = 0.f, xz = 0.f, yz = 0.f;
      const float Iv[3] = {I.x, I.y, I.z};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        xx += R.m[k] * Iv[k] * R.m[k]; yy += R.m[3 + k] * Iv[k] * R.m[3 + k]; zz += R.m[6 + k] * Iv[k] * R.m[6 + k];
        xy += R.m[k] * Iv[k] * R.m[3 + k]; xz += R.m[k] * Iv[k] * R.m[6 + k]; yz += R.m[3 + k] * Iv[k] * R.m[6 + k];
      }
      float r2 = dot(r, r);
      b_cinert[0] = xx + ms * (r2 - r.x * r.x); b_cinert[1] = yy + ms * (r2 - r.y * r.y); b_cinert[2] = zz + ms * (r2 - r.z * r.z);
      b_cinert[3] = xy - ms * r.x * r.y; b_cinert[4] = xz - ms * r.x * r.z; b_cinert[5] = yz - ms * r.y * r.z;
      b_cinert[6] = ms * r.x; b_cinert[7] = ms * r.y; b_cinert[8] = ms * r.z; b_cinert[9] = ms;
    } else {
#pragma unroll
      for (int k = 0; k < 10; k++) b_cinert[k] = 0.f;
    }
    // motion axes of the dof(s) owned by this lane (lane g == dof g)
    if (g < KD().nv) {
      int j = MI_(DOF_JNTID)[g], b = MI_(JNT_BODYID)[j], type = MI_(JNT_TYPE)[j], da = MI_(JNT_DOFADR)[j];
      V3 off = ld3(W + L.com + 3 * AUXI(dof_rootslot)[g]) - ld3(W + L.xanchor + 3 * j);
      V3 ang, lin;
      if (type == MM_JNT_HINGE) { ang = ld3(W + L.xaxis + 3 * j); lin = cross(ang, off); }
      else if (type == MM_JNT_SLIDE) { ang = v3(0.f, 0.f, 0.f); lin = ld3(W + L.xaxis + 3 * j); }
      else {
        int k = g - da;
        if (type == MM_JNT_FREE && k < 3) { ang = v3(0.f, 0.f, 0.f); lin = v3(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f); }
        else {
          if (type == MM_JNT_FREE) k -= 3;
          const float* R = W + L.xmat + 9 * b;
          ang = v3(R[k], R[3 + k], R[6 + k]); lin = cross(ang, off);
        }
      }
      d_cdof[0] = ang.x; d_cdof[1] = ang.y; d_cdof[2] = ang.z; d_cdof[3] = lin.x; d_cdof[4] = lin.y; d_cdof[5] = lin.z;
#pragma unroll
      for (int k = 0; k < 6; k++) W[L.cdof + 6 * g + k] = d_cdof[k];
    }
    GSYNC();
  }

  // ---------------------------------------------------------------- A2 tendons
  // Two sweeps.  (1) Path items, flattened over ALL tendons on the host (Aux.item_tab, 4 words each: {tendon | kind << 16,
  // site0 | site1 << 16, geom | (sidesite + 1) << 16, bits(1/divisor)}; kind 0 = site-site, 1 = site-sphere-site, 2 = site-cylinder-site, 3 = fixed-tendon
  // joint term) and sorted so that the expensive wrap items come first (a sweep of G lanes then executes one kind of item
  // instead of every lane walking its own tendon): lengths, and for wrap items the two tangent points + a wrapped flag into LDS.
  // (2) Jacobian entries, one lane per sparse-J entry (Aux.jent / jrec): the lane recomputes the end points of the segment(s)
  // that cross its dof and stores the entry.  The Jacobian used to be scattered from sweep (1) with LDS float atomics, a
  // data-dependent loop over each segment's dofs that ran to the longest list of the wave with two dependent LDS round trips
  // per turn: half of the stage.
  // Word offsets the sweeps need (LDS table bases, model sections, engine tables), read ONCE and pinned in SGPRs for the
  // duration of the stage: left to itself the compiler rematerialises each of them with an s_load + s_waitcnt lgkmcnt(0)
  // inside the loops (the wait also drains the LDS queue).
  struct TendonOff {
    int xpos, xmat, tenlen, tenj, xaxis, xanchor, com, cdof, qpos, wrapw;
    int site_body, site_pos, geom_body, geom_pos, geom_quat, geom_size, rootslot;
  };
  __device__ __forceinline__ V3 site_pos_o(const TendonOff& o, int s_) const {
    const int b = reinterpret_cast<const int*>(mb + o.site_body)[s_];
    return ld3(W + o.xpos + 3 * b) + mv(ldm(W + o.xmat + 9 * b), ld3(reinterpret_cast<const float*>(mb + o.site_pos) + 3 * s_));
  }

  __device__ __forceinline__ void tendon() {
    const auto& L = KL();
    TendonOff o;
    o.xpos = L.xpos; o.xmat = L.xmat; o.tenlen = L.tenlen; o.tenj = L.tenj; o.xaxis = L.xaxis; o.xanchor = L.xanchor;
    o.com = L.com; o.cdof = L.cdof; o.qpos = L.qpos; o.wrapw = L.wrapw;
    o.site_body = SECOFF_(SITE_BODYID); o.site_pos = SECOFF_(SITE_POS); o.geom_body = SECOFF_(GEOM_BODYID);
    o.geom_pos = SECOFF_(GEOM_POS); o.geom_quat = SECOFF_(GEOM_QUAT); o.geom_size = SECOFF_(GEOM_SIZE);
    o.rootslot = KX().dof_rootslot;
    int o_items = KX().item_tab, nitem = KX().nitem, o_jent = KX().jent, o_jrec = KX().jrec, d_ntenJ = KD().ntenJ;
    PIN_S(o.xpos); PIN_S(o.xmat); PIN_S(o.tenlen); PIN_S(o.tenj); PIN_S(o.xaxis); PIN_S(o.xanchor); PIN_S(o.site_body);
    PIN_S(o.site_pos); PIN_S(o.wrapw); PIN_S(o_items); PIN_S(nitem); PIN_S(o_jent); PIN_S(o_jrec); PIN_S(d_ntenJ);
    const int4* items = reinterpret_cast<const int4*>(mb + o_items);
    unsigned long long tt_ = (MM_STAGE_PROF && a.prof) ? clock64() : 0;
    for (int t = g; t < KD().ntendon; t += G) W[o.tenlen + t] = 0.f;
    GSYNC();
    for (int it = g; it < nitem; it += G) {
      const int4 I = items[it];   // [tendon | kind << 16, site0 | site1 << 16, geom | (sidesite + 1) << 16, bits(1 / divisor)]
      const int t = I.x & 0xffff, kind = I.x >> 16;
      const float inv_div = __int_as_float(I.w);
      if (kind == 3) {   // fixed tendon: coef * q_joint   [.., joint id, bits(coef), ..]
        atomicAdd(&W[o.tenlen + t], __int_as_float(I.z) * W[o.qpos + MI_(JNT_QPOSADR)[I.y]]);
        continue;
      }
      V3 p0 = site_pos_o(o, I.y & 0xffff), p1 = site_pos_o(o, (I.y >> 16) & 0xffff);
      float wlen = -1.f;
      V3 w0, w1;
      if (kind != 0) {
        const int gi = I.z & 0xffff, sideid = ((I.z >> 16) & 0xffff) - 1;
        V3 side = v3(0.f, 0.f, 0.f);
        if (sideid >= 0) side = site_pos_o(o, sideid);
        const int gb = reinterpret_cast<const int*>(mb + o.geom_body)[gi];
        const M3 A = ldm(W + o.xmat + 9 * gb), B = q2m(ldq(reinterpret_cast<const float*>(mb + o.geom_quat) + 4 * gi));
        const V3 gp = ld3(W + o.xpos + 3 * gb) + mv(A, ld3(reinterpret_cast<const float*>(mb + o.geom_pos) + 3 * gi));
        M3 R;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) R.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
        wlen = wrap_geom(w0, w1, p0, p1, gp, R, reinterpret_cast<const float*>(mb + o.geom_size)[3 * gi], kind == 2, sideid >= 0, side);
        float* ws = W + o.wrapw + 7 * it;
        if (wlen >= 0.f) { st3(ws, w0); st3(ws + 3, w1); }
        ws[6] = wlen < 0.f ? 0.f : 1.f;
      }
      if (wlen < 0.f) {
        V3 dif = p1 - p0;
        atomicAdd(&W[o.tenlen + t], sqrtf(dot(dif, dif)) * inv_div);
      } else {
        V3 d0 = w0 - p0, d1 = p1 - w1;
        float n0 = sqrtf(dot(d0, d0)), n1 = sqrtf(dot(d1, d1));
        atomicAdd(&W[o.tenlen + t], (n0 + wlen + n1) * inv_div);
      }
    }
    GSYNC();
    if (MM_STAGE_PROF && a.prof) { const unsigned long long t1_ = clock64(); pf[MM_STAGE_PROF ? PF_N_HBUILD : 0] += t1_ - tt_; tt_ = t1_; }   // (tools build: item sweep)
    const int* jent = reinterpret_cast<const int*>(mb + o_jent);
    const int4* jrow = reinterpret_cast<const int4*>(mb + o_jrec);
    int jn = g < d_ntenJ ? jent[g] : 0;
    for (int i = g; i < d_ntenJ; i += G) {
      const int r0 = jn & 0xffffff, nr = (jn >> 24) & 127;
      if (i + G < d_ntenJ) jn = jent[i + G];   // next sweep's row index: off this sweep's dependency chain
      float acc = 0.f;
      int e = 0;
      for (int r = r0; r < r0 + nr; r++) {
        // one row, then every load it addresses at once (both sites, both tangent points, the joint): two round trips per row
        const int4 ra = jrow[r];   // [entry | joint word << 16, site0 | site1 << 16, body0 | body1 << 8 | mode << 16 | eps << 20 | wrap slot << 22, bits(f)]
        const float f2 = __int_as_float(ra.w);
        e = ra.x & 0xffff;
        const int jw = ra.x >> 16, id = jw & 0xff, jk = (jw >> 8) & 3;
        const int mode = (ra.z >> 16) & 15;
        const int s0 = ra.y & 0xffff, s1 = (ra.y >> 16) & 0xffff, b0 = ra.z & 0xff, b1 = (ra.z >> 8) & 0xff;
        const float* ws = W + o.wrapw + 7 * ((ra.z >> 22) & 1023);
        const float wflag = ws[6];
        const V3 t0 = ld3(ws), t1 = ld3(ws + 3);
        const V3 q0 = ld3(W + o.xpos + 3 * b0) + mv(ldm(W + o.xmat + 9 * b0), ld3(reinterpret_cast<const float*>(mb + o.site_pos) + 3 * s0));
        const V3 q1 = ld3(W + o.xpos + 3 * b1) + mv(ldm(W + o.xmat + 9 * b1), ld3(reinterpret_cast<const float*>(mb + o.site_pos) + 3 * s1));
        // hinge: moment arm straight from the joint, u . (axis x (p - anchor)); slide: u . axis.  (Going through cdof -- motion
        // about the subtree COM -- adds and subtracts the COM offset: ~0.2 m against a 5 mm moment arm in the hand.)
        V3 ax, an, lin = v3(0.f, 0.f, 0.f);
        if (jk != 0) { ax = ld3(W + o.xaxis + 3 * id); an = ld3(W + o.xanchor + 3 * id); }
        else {   // ball / free dofs: motion axes about the subtree COM
          ax = ld3(W + o.cdof + 6 * id); lin = ld3(W + o.cdof + 6 * id + 3);
          an = ld3(W + o.com + 3 * reinterpret_cast<const int*>(mb + o.rootslot)[id]);
        }
        if (mode >= 6) { acc += mode == 6 ? f2 : 0.f; continue; }   // fixed tendon: the coefficient / nothing crosses the dof
        const bool wrapped = mode != 0 && wflag != 0.f;
        if (mode == 5 ? wrapped : (mode >= 3 && !wrapped)) continue;     // A only / B only, C only
        const bool tb = wrapped && (mode == 1 || mode == 3), tc = wrapped && (mode == 2 || mode == 4);
        const int ep = (ra.z >> (wrapped ? 21 : 20)) & 1;
        const V3 p0 = tc ? t1 : q0, p1 = tb ? t0 : q1;
        const V3 dif = p1 - p0;
        const float n = sqrtf(dot(dif, dif));
        const V3 u = n < MINVALF ? v3(f2, 0.f, 0.f) : (f2 / n) * dif;
        const V3 p = ep ? p1 : p0;
        const float val = jk == 2 ? dot(u, ax) : dot(u, lin + cross(ax, p - an));
        acc += ep ? val : -val;
      }
      W[o.tenj + e] = acc;
    }
    GSYNC();
    if (MM_STAGE_PROF && a.prof) pf[MM_STAGE_PROF ? PF_N_FACTOR : 0] += clock64() - tt_;   // (tools build: Jacobian-entry sweep)
  }

  // ------------------------------------------------------------- A7 joint-limit rows (one per lane)
  __device__ __forceinline__ void impedance(const float* si, const float* sr, float x, float diagApprox, float vel,
                                            float& D, float& aref) const {
    float dmin = clampf(si[0], 0.0001f, 0.9999f), dmax = clampf(si[1], 0.0001f, 0.9999f);
    float width = fmaxf(0.f, si[2]), mid = clampf(si[3], 0.0001f, 0.9999f), power = fmaxf(1.f, si[4]);
    float imp;
    if (width < MINVALF || dmin == dmax) imp = 0.5f * (dmin + dmax);
    else {
      float xa = fabsf(x) / width, y;
      if (xa >= 1.f) imp = dmax;
      else if (xa == 0.f) imp = dmin;
      else {
        if (power == 1.f) y = xa;
        else if (power == 2.f) y = xa <= mid ? xa * xa / mid : 1.f - (1.f - xa) * (1.f - xa) / (1.f - mid);
        else if (xa <= mid) y = powf(xa, power) / powf(mid, power - 1.f);
        else y = 1.f - powf(1.f - xa, power) / powf(1.f - mid, power - 1.f);
        imp = dmin + y * (dmax - dmin);
      }
    }
    float R = fmaxf(MINVALF, (1.f - imp) * diagApprox / imp);
    float K, B;
    if (sr[0] > 0.f) {
      float tc = fmaxf(sr[0], 2.f * KD().timestep), dr = sr[1];
      K = 1.f / fmaxf(MINVALF, dmax * dmax * tc * tc * dr * dr);
      B = 2.f / fmaxf(MINVALF, dmax * tc);
    } else { K = -sr[0] / fmaxf(MINVALF, dmax * dmax); B = -sr[1] / fmaxf(MINVALF, dmax); }
    D = 1.f / R;
    aref = -B * vel - K * imp * x;
  }

  // One potential limit row per JOINT, owned by lane j: a joint can violate only one side of its range at a
  // time (mm_model_create rejects ranges narrower than 2*margin).
  __device__ __forceinline__ void make_constraint() {
    if constexpr (GEN) { make_constraint_gen(); return; }
    const auto& L = KL();
    const int j = g;
    r_active = false; r_D = 0.f; r_aref = 0.f; r_dof = 0; r_sign = 1.f;
    if (j < KD().njnt) {
      int type = MI_(JNT_TYPE)[j];
      if (MI_(JNT_LIMITED)[j] && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) {
        r_dof = MI_(JNT_DOFADR)[j];
        float q = W[L.qpos + MI_(JNT_QPOSADR)[j]];
        float margin = MF_(JNT_MARGIN)[j];
        float dlo = q - MF_(JNT_RANGE)[2 * j], dhi = MF_(JNT_RANGE)[2 * j + 1] - q;
        float dist = dlo;
        if (!(dlo < margin) && dhi < margin) { dist = dhi; r_sign = -1.f; }
        if (dist < margin) {
          r_active = true;
          impedance(MF_(JNT_SOLIMP) + 5 * j, MF_(JNT_SOLREF) + 2 * j, dist - margin, MF_(DOF_INVWEIGHT0)[r_dof],
                    r_sign * W[L.qvel + r_dof], r_D, r_aref);
        }
      }
    }
    nefc = (int)(gsum<G>(r_active ? 1.f : 0.f) + 0.5f);
  }

  // value of the limit row of the joint that owns dof g (0 for dofs that are not a hinge/slide joint's dof)
  __device__ __forceinline__ float rows_to_dof(float val) const {
    const float v = sh<G>(val, c_rowj);
    return c_rowj_mine ? v : 0.f;
  }

  // tendon velocities J qvel (sparse rows)
  __device__ __forceinline__ void tendon_velocity() {
    const auto& L = KL();
    int o_qvel = L.qvel;
    {
      int s_ja = SECOFF_(TENJ_ADR), s_jd = SECOFF_(TENJ_DOF), o_tj = L.tenj, o_qv = o_qvel, o_tv = L.tenvel;
      PIN_S(s_ja); PIN_S(s_jd); PIN_S(o_tj); PIN_S(o_qv); PIN_S(o_tv);
      for (int t = g; t < KD().ntendon; t += G) {
        float s = 0.f;
        const int e0 = reinterpret_cast<const int*>(mb + s_ja)[t], e1 = reinterpret_cast<const int*>(mb + s_ja)[t + 1];
        for (int e = e0; e < e1; e++) s += W[o_tj + e] * W[o_qv + reinterpret_cast<const int*>(mb + s_jd)[e]];
        W[o_tv + t] = s;
      }
    }
  }

  // Subtree sums of a K-vector per body, in place in an LDS table [nbody][K] (composite inertias, RNE forces): S[b] = V[b] +
  // sum over the children c of S[c].  The body tree is cut into chains (maximal unbranched paths; bodies of a chain have
  // consecutive ids in MuJoCo's depth-first order -- the host checks); the lane of a chain's top body walks its chain from the
  // bottom up in registers, starting from its bottom body's value plus the finished totals of the chains hanging off it.  One
  // step per level of the CHAIN tree (3 for the hand and the leg) instead of one per level of the body tree (9 / 12), and no LDS
  // float atomics.  Falls back to the level-by-level atomic sweep when the host could not build the chains (bchain_nlevel = 0).
  template <int K>
  __device__ __forceinline__ void subtree_sum(int o_tab) {
    const int ncl = KD().bchain_nlevel & 15, maxch = (KD().bchain_nlevel >> 4) & 15, maxlen = KD().bchain_nlevel >> 8;
    int bottom = 0, lv = -1, nch = 0;
    unsigned ch_lo = 0u, ch_hi = 0u;
    if (g > 0 && g < KD().nbody) {
      const int* bc = AUXI(body_chain) + 3 * g;
      const int w = bc[0];
      if (w >= 0) { bottom = w & 255; lv = (w >> 8) & 15; nch = (w >> 12) & 15; ch_lo = (unsigned)bc[1]; ch_hi = (unsigned)bc[2]; }
    }
    for (int cl = ncl - 1; cl >= 0; cl--) {
      if (lv == cl) {
        float acc[K];
#pragma unroll
        for (int k = 0; k < K; k++) acc[k] = W[o_tab + K * bottom + k];
        // wave-uniform trip counts (the most children / the longest chain of the model) with per-lane predicates: data-dependent
        // per-lane loops here run into a backend error ("illegal VGPR to SGPR copy") in some instantiations
        for (int c = 0; c < maxch; c++) {
          if (c < nch) {
            const int ct = (int)((c < 4 ? ch_lo >> (8 * c) : ch_hi >> (8 * (c - 4))) & 255u);
#pragma unroll
            for (int k = 0; k < K; k++) acc[k] += W[o_tab + K * ct + k];
          }
        }
        if (nch > 0)
#pragma unroll
          for (int k = 0; k < K; k++) W[o_tab + K * bottom + k] = acc[k];
        for (int st = 1; st < maxlen; st++) {
          const int b = bottom - st;
          if (b >= g) {
#pragma unroll
            for (int k = 0; k < K; k++) { acc[k] += W[o_tab + K * b + k]; W[o_tab + K * b + k] = acc[k]; }
          }
        }
      }
      GSYNC();
    }
  }

  // ----------------------------------------------------- A5 velocity stage + bias forces
  __device__ __forceinline__ void velocity_bias() {
    // offsets read once and pinned in SGPRs for this stage (see PIN_S)
    int o_cdof = KL().cdof; PIN_S(o_cdof); int o_u1 = KL().u1; PIN_S(o_u1); int o_qvel = KL().qvel; PIN_S(o_qvel); int s_DOF_BODYID = SECOFF_(DOF_BODYID); PIN_S(s_DOF_BODYID); int d_nlevel_ = KD().nlevel; PIN_S(d_nlevel_); int d_nbody_ = KD().nbody; PIN_S(d_nbody_); int d_nv_ = KD().nv; PIN_S(d_nv_);
    const auto& L = KL();
    const int nb = d_nbody_;
    if (!(TW && a.two_wave)) tendon_velocity();
    // Forward pass by pointer jumping (see kinematics): cvel of a body is the SUM of cdof * qvel over the dofs of its
    // ancestors and itself (everything is expressed about the subtree COM: no frame change along the chain), so it is a
    // prefix sum over the chain: ceil(log2(depth)) rounds of "add what my pointer holds, point where it points".  The bias
    // acceleration cacc is the same kind of sum of (cvel before the dof) x cdof * qvel, plus -gravity at the root, and is
    // summed the same way once cvel is final.
    float cv[6], ca[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { cv[k] = 0.f; ca[k] = 0.f; }
    const bool isb = g > 0 && g < nb;
    int nround = 0;
    for (int s_ = 1; s_ < d_nlevel_; s_ <<= 1) nround++;
    int* UP = reinterpret_cast<int*>(W + o_u1 + CVS * nb);   // pointer scratch behind the (cvel, cacc) slots: u1 holds >= (CVS + 1) nbody words
    const int* JP = AUXI(jnt_pack);
    const int j0_ = c_ja > 0 ? c_ja : 0;
    if (isb) {      // own dofs: local velocity contribution
      int pk_ = JP[2 * j0_];
      for (int i = 0; i < c_jn; i++) {
        int type = pk_ & 15, da = (pk_ >> 4) & 1023;
        pk_ = JP[2 * (i + 1 < c_jn ? c_ja + i + 1 : c_ja + i)];     // (the next joint's word is in flight while this one's dofs are read)
        const int nd = type == MM_JNT_FREE ? 6 : (type == MM_JNT_BALL ? 3 : 1);
        for (int d3 = 0; d3 < nd; d3++) {
          const float qv = W[o_qvel + da + d3];
#pragma unroll
          for (int k = 0; k < 6; k++) cv[k] += W[o_cdof + 6 * (da + d3) + k] * qv;
        }
      }
    }
    float own[6];
#pragma unroll
    for (int k = 0; k < 6; k++) own[k] = cv[k];
    {
      int up = isb ? b_parent : 0;
      for (int r = 0; r < nround; r++) {
        if (g < nb) {
#pragma unroll
          for (int k = 0; k < 6; k++) W[o_u1 + CVS * g + k] = cv[k];
          UP[g] = up;
        }
        GSYNC();
        if (up > 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) cv[k] += W[o_u1 + CVS * up + k];
          up = UP[up];
        }
        GSYNC();
      }
    }
    // local acceleration contribution: walk the own dofs again with the running velocity (parent's cvel first)
    if (isb) {
      float run[6];
#pragma unroll
      for (int k = 0; k < 6; k++) run[k] = cv[k] - own[k];
      int pk_ = JP[2 * j0_];
      for (int i = 0; i < c_jn; i++) {
        int type = pk_ & 15, da = (pk_ >> 4) & 1023;
        pk_ = JP[2 * (i + 1 < c_jn ? c_ja + i + 1 : c_ja + i)];
        if (type == MM_JNT_FREE) {     // translational dofs: cdof_dot = 0, they only move the running velocity
          for (int d3 = 0; d3 < 3; d3++) {
            const float qv = W[o_qvel + da + d3];
#pragma unroll
            for (int k = 0; k < 6; k++) run[k] += W[o_cdof + 6 * (da + d3) + k] * qv;
          }
          da += 3;
          type = MM_JNT_BALL;
        }
        const int nd = type == MM_JNT_BALL ? 3 : 1;
        float base[6];
#pragma unroll
        for (int k = 0; k < 6; k++) base[k] = run[k];
        for (int d3 = 0; d3 < nd; d3++) {
          float cd[6], cdd[6];
#pragma unroll
          for (int k = 0; k < 6; k++) cd[k] = W[o_cdof + 6 * (da + d3) + k];
          cross_motion(cdd, base, cd);
          const float qv = W[o_qvel + da + d3];
#pragma unroll
          for (int k = 0; k < 6; k++) { run[k] += cd[k] * qv; ca[k] += cdd[k] * qv; }
        }
      }
    }
    {
      int up = isb ? b_parent : 0;
      for (int r = 0; r < nround; r++) {
        if (g < nb) {
#pragma unroll
          for (int k = 0; k < 6; k++) W[o_u1 + CVS * g + 6 + k] = ca[k];
          UP[g] = up;
        }
        GSYNC();
        if (up > 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) ca[k] += W[o_u1 + CVS * up + 6 + k];
          up = UP[up];
        }
        GSYNC();
      }
    }
    if (g < nb) { ca[3] -= KD().gx; ca[4] -= KD().gy; ca[5] -= KD().gz; }   // the world body's cacc = -gravity reaches everybody
#pragma unroll
    for (int k = 0; k < 6; k++) b_cvel[k] = cv[k];
    // cfrc_body = I*cacc + cvel x* (I*cvel)
    float cf[6];
    {
      float Ia[6], Iv[6], x[6];
      inert_mul(Ia, b_cinert, ca); inert_mul(Iv, b_cinert, cv); cross_force(x, cv, Iv);
#pragma unroll
      for (int k = 0; k < 6; k++) cf[k] = (g > 0 && g < nb) ? Ia[k] + x[k] : 0.f;
    }
    // backward accumulation through LDS (u1 region now holds cfrc[6*nbody]); deepest level first
    GSYNC();
    if (KD().bchain_nlevel > 0) {
      if (g < nb)
#pragma unroll
        for (int k = 0; k < 6; k++) W[o_u1 + 6 * g + k] = cf[k];
      GSYNC();
      subtree_sum<6>(o_u1);
    } else {
      if (g < nb)
#pragma unroll
        for (int k = 0; k < 6; k++) W[o_u1 + 6 * g + k] = 0.f;
      GSYNC();
      for (int lv = d_nlevel_; lv >= 1; lv--) {
        if (b_depth == lv) {
#pragma unroll
          for (int k = 0; k < 6; k++) {
            float tot = cf[k] + W[o_u1 + 6 * g + k];
            W[o_u1 + 6 * g + k] = tot;
            if (b_parent > 0) atomicAdd(&W[o_u1 + 6 * b_parent + k], tot);
          }
        }
        GSYNC();
      }
    }
    d_bias = 0.f;
    if (g < d_nv_) {
      int b = AI_(s_DOF_BODYID)[g];
#pragma unroll
      for (int k = 0; k < 6; k++) d_bias += d_cdof[k] * W[o_u1 + 6 * b + k];
    }
    GSYNC();
  }

  // ---------------------------------------------------------------- A4 CRB -> dense M rows
  __device__ __forceinline__ void crb() {
    // offsets read once and pinned in SGPRs for this stage (see PIN_S)
    int o_cdof = KL().cdof; PIN_S(o_cdof); int o_u1 = KL().u1; PIN_S(o_u1); int o_crb = KL().crb; PIN_S(o_crb); int s_DOF_PARENTID = SECOFF_(DOF_PARENTID); PIN_S(s_DOF_PARENTID); int s_DOF_BODYID = SECOFF_(DOF_BODYID); PIN_S(s_DOF_BODYID); int s_DOF_ARMATURE = SECOFF_(DOF_ARMATURE); PIN_S(s_DOF_ARMATURE); int d_nv_ = KD().nv; PIN_S(d_nv_); int d_nbody_ = KD().nbody; PIN_S(d_nbody_);
    const auto& L = KL();
    const int nb = d_nbody_, nv = d_nv_;
    if (g < nb)
#pragma unroll
      for (int k = 0; k < 10; k++) W[o_crb + 10 * g + k] = b_cinert[k];
    // zero the dense tile (u1 region; cfrc is dead now) and put 1 on the padded diagonal
    if constexpr (!SP)
      for (int e = g; e < NVP * TD; e += G) W[o_u1 + e] = 0.f;
    GSYNC();
    if (KD().bchain_nlevel > 0) subtree_sum<10>(o_crb);
    else
      for (int lv = KD().nlevel; lv >= 2; lv--) {
        if (b_depth == lv && b_parent > 0)
#pragma unroll
          for (int k = 0; k < 10; k++) atomicAdd(&W[o_crb + 10 * b_parent + k], W[o_crb + 10 * g + k]);
        GSYNC();
      }
    if constexpr (SP) {
      // M[g][anc_d] = cdof_anc . (Ic_body(g) cdof_g): the ancestors' motion axes are independent LDS gathers
      if (g < nv) {
        float I[10], buf[6];
        int b = AI_(s_DOF_BODYID)[g];
#pragma unroll
        for (int k = 0; k < 10; k++) I[k] = W[o_crb + 10 * b + k];
        inert_mul(buf, I, d_cdof);
        float diag = AF_(s_DOF_ARMATURE)[g];
#pragma unroll
        for (int k = 0; k < 6; k++) diag += d_cdof[k] * buf[k];
        Md = diag;
        sp_crb_entry<0>(buf, o_cdof);
      }
      GSYNC();
      return;
    }
    if (g < nv) {
      float I[10], buf[6];
      int b = AI_(s_DOF_BODYID)[g];
#pragma unroll
      for (int k = 0; k < 10; k++) I[k] = W[o_crb + 10 * b + k];
      inert_mul(buf, I, d_cdof);
      const int* dpar = AI_(s_DOF_PARENTID);
      // ancestor walk with the parent pointer fetched one step ahead: the motion axes of ancestor j and the pointer two above it
      // are in flight together (pointer -> axes -> pointer was two dependent round trips per level, 17 levels for the leg)
      const float arm = AF_(s_DOF_ARMATURE)[g];
      int j = g, jn = dpar[g];
      while (j >= 0) {
        float cj[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cj[k] = W[o_cdof + 6 * j + k];
        int jnn = dpar[jn >= 0 ? jn : 0];
        jnn = jn >= 0 ? jnn : -1;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) s += cj[k] * buf[k];
        if (j == g) s += arm;
        W[o_u1 + g * TD + j] = s;
        W[o_u1 + j * TD + g] = s;
        j = jn; jn = jnn;
      }
    } else if (g < NVP) {
      W[o_u1 + g * TD + g] = 1.f;
    }
    GSYNC();
    if (g < NVP) {
#pragma unroll
      for (int k4 = 0; k4 < NVP / 4; k4++) {
        const float4 r = *reinterpret_cast<const float4*>(W + o_u1 + g * TD + 4 * k4);
        Mrow[4 * k4] = r.x; Mrow[4 * k4 + 1] = r.y; Mrow[4 * k4 + 2] = r.z; Mrow[4 * k4 + 3] = r.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NVP; k++) Mrow[k] = 0.f;
    }
    GSYNC();
  }

  template <int E_>
  __device__ __forceinline__ void sp_crb_entry(const float (&buf)[6], int o_cdof) {
    if constexpr (E_ < SD) {
      float s = 0.f;
      if constexpr (E_ < SD - 1) {
        if (E_ < d_depth) {
          const int j = anc<E_>();
#pragma unroll
          for (int k = 0; k < 6; k++) s += W[o_cdof + 6 * j + k] * buf[k];
        }
      }
      Ms[E_] = s;
      sp_crb_entry<E_ + 1>(buf, o_cdof);
    }
  }

  // ---- tree-sparse L'DL (MuJoCo's mj_factorI / mj_solveLD order: leaves first, no fill-in) fused with its one solve.
  // Every factorisation in an SP kernel is followed by exactly one solve, so the elimination carries the right-hand side.
  // The dof tree is cut into SEGMENTS, maximal unbranched chains (hand: the wrist chain and five finger chains).  One lane --
  // the lane of the segment's top dof -- eliminates the whole segment in registers: it assembles the frontal matrix over the
  // depths 0 .. bottom (rows of its own dofs, published by their lanes, plus the update matrices of its child segments),
  // runs the dense L'DL pivots of its own depths with compile-time indices, keeps L for the back substitution and publishes
  // the update matrix over its ancestors for the parent segment.  Sequential steps = levels of the SEGMENT tree (2 for the
  // hand), ~350 instructions per factor + solve against ~1200 with one lane per dof and one step per tree level, and ~1400 for
  // the dense form (24 pivots with a broadcast and an LDS round trip each, 48 substitution steps).  No LDS float atomics:
  // gfx950 executes a ds_add_f32 at ~3 cycles per ACTIVE LANE, CU-wide (tools/micro/lds_atomic_bench.hip: 192 cycles for 64
  // lanes against 8 for a ds_write_b32).
  static constexpr int tri(int d) { return d * (d + 1) / 2; }
  template <int D_>
  __device__ __forceinline__ static int sg_path(const SegLane& q) {
    return (int)(D_ < 4 ? (q.path_lo >> (8 * D_)) & 255u : (q.path_hi >> (8 * (D_ - 4))) & 255u);
  }
  __device__ __forceinline__ static unsigned byte_of(unsigned w0, unsigned w1, int c) {
    return ((c < 4 ? w0 >> (8 * c) : w1 >> (8 * (c - 4))) & 255u);
  }
  // (t, b): depth range of the segments of the level at hand -- the same for every segment of one level of an SP model (the
  // host routes other trees to the general-row kernels), so these are scalar branches around straight-line code
  template <int D_>
  __device__ __forceinline__ void sg_load_rows(const SegLane& q, const float* T, float (&F)[36], float (&r)[SD], int t, int b) const {
    if constexpr (D_ < SD) {
      if (D_ >= t && D_ <= b) {
        const float* P = T + sg_path<D_>(q) * TS;
        const float4 p0 = *reinterpret_cast<const float4*>(P);
        const float v0[4] = {p0.x, p0.y, p0.z, p0.w};
#pragma unroll
        for (int e = 0; e <= (D_ < 3 ? D_ : 3); e++) F[tri(D_) + e] = v0[e];
        if constexpr (D_ >= 4) {
          const float4 p1 = *reinterpret_cast<const float4*>(P + 4);
          const float v1[4] = {p1.x, p1.y, p1.z, p1.w};
#pragma unroll
          for (int e = 4; e <= D_; e++) F[tri(D_) + e] = v1[e - 4];
        }
        r[D_] = P[8];
      }
      sg_load_rows<D_ + 1>(q, T, F, r, t, b);
    }
  }
  template <int D_>
  __device__ __forceinline__ void sg_pivots(float (&F)[36], float (&r)[SD], float (&iv)[SD], int t, int b) const {
    if constexpr (D_ >= 0) {
      if (D_ >= t && D_ <= b) {
        const float inv = 1.f / fmaxf(F[tri(D_) + D_], MINVALF);
        iv[D_] = inv;
#pragma unroll
        for (int e = D_ - 1; e >= 0; e--) {       // descending: F[D][e2], e2 <= e, is still unscaled when row e needs it
          const float tt = F[tri(D_) + e] * inv;  // L[D][e]
#pragma unroll
          for (int e2 = 0; e2 <= e; e2++) F[tri(e) + e2] -= tt * F[tri(D_) + e2];
          r[e] -= tt * r[D_];
          F[tri(D_) + e] = tt;
        }
      }
      sg_pivots<D_ - 1>(F, r, iv, t, b);
    }
  }
  template <int D_>
  __device__ __forceinline__ void sg_anc_x(const SegLane& q, const float* X, float (&xa)[SD], int t) const {
    if constexpr (D_ < SD - 1) {
      xa[D_] = X[sg_path<D_>(q)];   // (unconditional: bytes beyond the path are 0 = dof 0)
      sg_anc_x<D_ + 1>(q, X, xa, t);
    } else xa[D_] = 0.f;
  }
  template <int D_>
  __device__ __forceinline__ void sg_back(const SegLane& q, float* X, const float (&xa)[SD], const float (&F)[36], const float (&r)[SD], const float (&iv)[SD], float (&x)[SD], int t, int b) const {
    if constexpr (D_ < SD) {
      if (D_ < t) x[D_] = xa[D_];
      else if (D_ <= b) {
        float v = r[D_] * iv[D_];
#pragma unroll
        for (int e = 0; e < D_; e++) v -= F[tri(D_) + e] * x[e];
        x[D_] = v;
        X[sg_path<D_>(q)] = v;
      }
      sg_back<D_ + 1>(q, X, xa, F, r, iv, x, t, b);
    }
  }
  // add the update matrices of the child segments (nq quads of the triangle + the rhs); slots of absent children read zeros.
  // SHALLOW (parents no deeper than depth 2, the hand's wrist): two quads + rhs in flight per child; otherwise one quad at a
  // time -- the 36-word front and a whole child matrix in flight do not fit the register file next to the engine's state.
  template <bool SHALLOW>
  __device__ __forceinline__ void sg_children(const SegLane& q, const float* U, int mch, int nq, int zslot, float (&F)[36], float (&r)[SD]) const {
    for (int c = 0; c < mch; c++) {
      const unsigned id = byte_of(q.ch_lo, q.ch_hi, c);
      const float* Uc = U + (id == 255u ? zslot : (int)id) * 36;
      if constexpr (SHALLOW) {
        const float4 u0 = *reinterpret_cast<const float4*>(Uc), u1 = *reinterpret_cast<const float4*>(Uc + 4);
        const float4 r0 = *reinterpret_cast<const float4*>(Uc + 28);
        F[0] += u0.x; F[1] += u0.y; F[2] += u0.z; F[3] += u0.w; F[4] += u1.x; F[5] += u1.y;
        r[0] += r0.x; r[1] += r0.y; r[2] += r0.z;
      } else {
#pragma unroll
        for (int k4 = 0; k4 < 7; k4++)
          if (k4 < nq) {
            const float4 u = *reinterpret_cast<const float4*>(Uc + 4 * k4);
            F[4 * k4] += u.x; F[4 * k4 + 1] += u.y; F[4 * k4 + 2] += u.z; F[4 * k4 + 3] += u.w;
          }
        const float4 r0 = *reinterpret_cast<const float4*>(Uc + 28), r1 = *reinterpret_cast<const float4*>(Uc + 32);
        r[0] += r0.x; r[1] += r0.y; r[2] += r0.z; r[3] += r0.w; r[4] += r1.x; r[5] += r1.y; r[6] += r1.z; r[7] += r1.w;
      }
    }
  }
  template <int NQ>
  __device__ __forceinline__ void sg_publish(float* Us, const float (&F)[36], const float (&r)[SD]) const {
#pragma unroll
    for (int q = 0; q < NQ; q++) *reinterpret_cast<float4*>(Us + 4 * q) = make_float4(F[4 * q], F[4 * q + 1], F[4 * q + 2], F[4 * q + 3]);
    *reinterpret_cast<float4*>(Us + 28) = make_float4(r[0], r[1], r[2], r[3]);
    *reinterpret_cast<float4*>(Us + 32) = make_float4(r[4], r[5], r[6], r[7]);
  }
  // x = A^-1 rhs; P = this lane's row of A by absolute depth (diagonal at the dof's depth), q = its segment data
  __device__ __forceinline__ float sp_solve_rows(const SegLane& q, const float (&P)[SD], float rhs) {
    float* T = W + KL().u1;            // published rows [NVP][TS]: row by absolute depth, rhs at [8]
    float* X = T + NVP * TS;           // solution by dof
    float* U = T + KD().seg_u;         // update matrices [segment][36]: lower triangle over depths (28 words), rhs (8 words)
    const int nsl = KD().seg_nlevel, zslot = KD().seg_zero;
    const unsigned info_lo = (unsigned)KD().seg_lvinfo[0], info_hi = (unsigned)KD().seg_lvinfo[1];
    const unsigned tb_lo = (unsigned)KD().seg_lvtb[0], tb_hi = (unsigned)KD().seg_lvtb[1];
    if (q.depth >= 0) {
      float* Pg = T + g * TS;
      *reinterpret_cast<float4*>(Pg) = make_float4(P[0], P[1], P[2], P[3]);
      *reinterpret_cast<float4*>(Pg + 4) = make_float4(P[4], P[5], P[6], P[7]);
      Pg[8] = rhs;
    }
    if (g < 9) *reinterpret_cast<float4*>(U + zslot * 36 + 4 * g) = make_float4(0.f, 0.f, 0.f, 0.f);
    GSYNC();
    float F[36], r[SD], iv[SD];
#pragma unroll
    for (int k = 0; k < 36; k++) F[k] = 0.f;
#pragma unroll
    for (int k = 0; k < SD; k++) { r[k] = 0.f; iv[k] = 1.f; }
    for (int sl = nsl - 1; sl >= 0; sl--) {
      const unsigned tb = byte_of(tb_lo, tb_hi, sl);
      const int t = (int)(tb & 15u), b = (int)(tb >> 4);
      if (q.lv == sl) {
        sg_load_rows<0>(q, T, F, r, t, b);
        // child segments: their update matrices cover the depths 0 .. b, a linear prefix of the triangle
        const int mch = (int)(byte_of(info_lo, info_hi, sl) & 15u);
        if (mch > 0) {
          if (b <= 2) sg_children<true>(q, U, mch, 2, zslot, F, r);
          else sg_children<false>(q, U, mch, (tri(b + 1) + 3) >> 2, zslot, F, r);
        }
        sg_pivots<SD - 1>(F, r, iv, t, b);
        if (t > 0) {   // update matrix over the ancestors: depths 0 .. t - 1
          float* Us = U + q.id * 36;
          if (t <= 3) sg_publish<2>(Us, F, r);
          else sg_publish<7>(Us, F, r);
        }
      }
      if (sl > 0) GSYNC();
    }
    float x[SD];
#pragma unroll
    for (int k = 0; k < SD; k++) x[k] = 0.f;
    for (int sl = 0; sl < nsl; sl++) {
      const unsigned tb = byte_of(tb_lo, tb_hi, sl);
      if (q.lv == sl) {
        // the ancestors' solution entries (depths < t), all requested at once: one uniform branch per depth with a ds_read in it
        // was one LDS round trip per depth (path bytes beyond the segment's top point at dof 0: harmless reads)
        const int t_ = (int)(tb & 15u);
        float xa[SD];
        sg_anc_x<0>(q, X, xa, t_);
        sg_back<0>(q, X, xa, F, r, iv, x, t_, (int)(tb >> 4));
      }
      GSYNC();
    }
    const float out = q.depth >= 0 ? X[g] : 0.f;
    GSYNC();
    return out;
  }
  // x = (A + diag(dadd))^-1 rhs for the matrix whose sparse rows are (Ms, Md): the limit-rows-only kernels
  __device__ __forceinline__ float sp_factor_solve(float dadd, float rhs) {
    float P[SD];
#pragma unroll
    for (int e = 0; e < SD; e++) P[e] = e == d_depth ? Md + dadd : Ms[e];
    return sp_solve_rows(sgl, P, rhs);
  }
  // the same for the general-row kernels (solve0 and the implicit-damping solve of the Euler step, whose matrices have M's
  // pattern; the Newton Hessian M + J'DJ does not).  These kernels hold M as dense rows (Mrow); the row by absolute ancestor
  // depth is picked out of it through the (dead) dense LDS tile, and the lane's segment data is re-read from the model tables
  // instead of held in registers across the constraint stages.
  __device__ __forceinline__ float spg_factor_solve(float dadd, float rhs) {
    const SegLane q = seg_lane_load();
    float* T = W + KL().u1;
    const int row = g < NVP ? g : 0;
    if (g < NVP)
#pragma unroll
      for (int k4 = 0; k4 < NVP / 4; k4++)
        *reinterpret_cast<float4*>(T + row * TD + 4 * k4) = make_float4(Mrow[4 * k4], Mrow[4 * k4 + 1], Mrow[4 * k4 + 2], Mrow[4 * k4 + 3]);
    float P[SD];
    {
      const int* pa = AUXI(dof_seg) + 6 * (q.depth >= 0 ? g : 0);
      // the dof's own path: ancestors by depth, then itself (words 1, 2 of an owner row describe the segment's BOTTOM dof;
      // a dof's own ancestors are the same bytes up to its depth)
      const unsigned a_lo = (unsigned)AUXI(dof_anc)[2 * (q.depth >= 0 ? g : 0)], a_hi = (unsigned)AUXI(dof_anc)[2 * (q.depth >= 0 ? g : 0) + 1];
      (void)pa;
#pragma unroll
      for (int e = 0; e < SD; e++) {
        const int col = e == q.depth ? g : (int)byte_of(a_lo, a_hi, e);
        const float v = T[row * TD + (e <= q.depth ? col : row)];
        P[e] = e < q.depth ? v : (e == q.depth ? v + dadd : 0.f);
      }
    }
    GSYNC();   // the rows are read: the tile region becomes the sparse solve's work area
    return sp_solve_rows(q, P, rhs);
  }
  // y = M x: the diagonal and ancestor entries are this lane's row; the descendant entries M[k][g] x_k are published by the
  // descendants (their row times their x) and summed through the dof's descendant list
  __device__ __forceinline__ float sp_mul_m(float x) const {
    float* Q = W + KL().u1;            // [NVP][SD]
    float* X = Q + NVP * TS;
    const int di = d_depth;
    if (di >= 0) {
      *reinterpret_cast<float4*>(Q + g * SD) = make_float4(Ms[0] * x, Ms[1] * x, Ms[2] * x, Ms[3] * x);
      *reinterpret_cast<float4*>(Q + g * SD + 4) = make_float4(Ms[4] * x, Ms[5] * x, Ms[6] * x, Ms[7] * x);
      X[g] = x;
    }
    const int nw = KD().desc_words;
    unsigned dw[8];
    {
      const int* dt = AUXI(dof_desc) + 8 * (di >= 0 ? g : 0);
#pragma unroll
      for (int q = 0; q < 8; q++) dw[q] = (q < nw && di >= 0) ? (unsigned)dt[q] : 0xffffffffu;
    }
    GSYNC();
    float y = 0.f;
    if (di >= 0) {
      // Ms is zero beyond the lane's depth and the unused ancestor bytes point at dof 0: no predicates
      y = Md * x + Ms[0] * X[anc<0>()] + Ms[1] * X[anc<1>()] + Ms[2] * X[anc<2>()] + Ms[3] * X[anc<3>()]
          + Ms[4] * X[anc<4>()] + Ms[5] * X[anc<5>()] + Ms[6] * X[anc<6>()];
#pragma unroll
      for (int q = 0; q < 8; q++)
        if (q < nw) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const unsigned k = (dw[q] >> (8 * c)) & 255u;
            v[c] = Q[(k != 255u ? (int)k : g) * SD + di];
            if (k == 255u) v[c] = 0.f;
          }
          y += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
    GSYNC();
    return y;
  }
  template <int E_>
  __device__ __forceinline__ void sp_dense_entry(float* T) const {
    if constexpr (E_ < SD - 1) {
      if (E_ < d_depth) { const int ja = anc<E_>(); T[g * TD + ja] = Ms[E_]; T[ja * TD + g] = Ms[E_]; }
      sp_dense_entry<E_ + 1>(T);
    }
  }
  // tests only: M as a dense symmetric NVP x NVP tile in the u1 region
  __device__ __forceinline__ void sp_dense_tile() const {
    float* T = W + KL().u1;
    for (int e = g; e < NVP * TD; e += G) T[e] = 0.f;
    GSYNC();
    if (d_depth >= 0) { T[g * TD + g] = Md; sp_dense_entry<0>(T); }
    GSYNC();
  }
  // (A + diag(dadd))^-1 rhs: the one entry point of every factor + solve pair
  __device__ __forceinline__ float factor_solve(float dadd, float rhs) {
    if constexpr (SP) return sp_factor_solve(dadd, rhs);
    else {
      if constexpr (GEN && MM_SPARSE_LDL && MM_SPARSE_GEN && NVP >= 8 && INTEG != 2) { if (MM_SPARSE_GEN == 2 || KD().seg_nlevel > 0) return spg_factor_solve(dadd, rhs); }
      factor(dadd); return solve(rhs);
    }
  }

  // dense Cholesky H = L L' with lane i holding row i; `dadd` is added to this lane's diagonal element.
  // Right-looking form: after column j is scaled, the updates of the trailing columns are independent FMAs
  // (instruction-level parallelism) instead of one serial dot-product chain per column.
  // Leaves Lrow (L[g][k]) and d_dinv (1/L[g][g]) in registers and L in the LDS tile (for the L' solve).
  __device__ __forceinline__ void factor(float dadd) {
    if constexpr (LEFT_LOOKING) { factor_core<true>(Mrow, dadd); return; }   // left-looking: reads M[g][j] once, no copy
    float A[NVP];
    int gq = g;
    asm volatile("" : "+v"(gq));   // (see factor_core: lane masks are not to be shared between the inlined copies)
#pragma unroll
    for (int k = 0; k < NVP; k++) A[k] = Mrow[k] + (k == gq ? dadd : 0.f);
    factor_core<false>(A, 0.f);
  }
  template <bool DIAG>
  __device__ __forceinline__ void factor_core(float (&A)[NVP], float dadd = 0.f) {
    const auto& L = KL();
    if constexpr (LEFT_LOOKING) {
      // Left-looking form for groups narrower than the wave.  A cross-lane broadcast costs ~5 issue slots there (two
      // v_readlane + v_mov + v_cndmask + hazard nops), and the right-looking update needs NVP^2/2 of them.  Here row j of L
      // comes from the LDS tile instead (one 128-bit load per four entries; the tile is written column by column as the
      // factor proceeds, LDS ops of a wave execute in order) and only the pivot is broadcast: NVP broadcasts in total.
      float* T = W + o_tile;
      const int row = g < NVP ? g : 0;
      int gq = g;                      // (opaque per call: see the right-looking branch below)
      asm volatile("" : "+v"(gq));
#pragma unroll
      for (int j = 0; j < NVP; j++) {
        float s = A[j];
        if constexpr (DIAG) s += (gq == j) ? dadd : 0.f;
#pragma unroll
        for (int k4 = 0; k4 < (j + 3) / 4; k4++) {
          const float4 r = *reinterpret_cast<const float4*>(T + j * TD + 4 * k4);
          if (4 * k4 + 0 < j) s -= Lrow[4 * k4 + 0] * r.x;
          if (4 * k4 + 1 < j) s -= Lrow[4 * k4 + 1] * r.y;
          if (4 * k4 + 2 < j) s -= Lrow[4 * k4 + 2] * r.z;
          if (4 * k4 + 3 < j) s -= Lrow[4 * k4 + 3] * r.w;
        }
        const float piv = bc<G>(s, j);
        const float inv = __frsqrt_rn(fmaxf(piv, MINVALF));
        const float lj = (gq > j) ? s * inv : 0.f;    // STRICTLY lower: the diagonal lives in d_dinv (see scale_rows)
        Lrow[j] = lj;
        if (gq == j) d_dinv = inv;
        if (g < NVP) T[row * TD + j] = lj;
      }
      if (g >= NVP) d_dinv = 1.f;
      GSYNC();
      scale_rows();
      return;
    }
    // The lane index the 2 NVP comparisons below use is made opaque per call: the factorisation is inlined several times per
    // pipeline copy, the compiler shared the (g > j), (g == j) lane masks between the copies, kept 72 of them alive in SGPR
    // pairs across the stages in between, spilled them -- and every use became two v_readlane + s_nop (690 restores in the leg
    // kernel, ~170 per factorisation of 1476 instructions).
    int gq = g;
    asm volatile("" : "+v"(gq));
#pragma unroll
    for (int j = 0; j < NVP; j++) {
      float piv = bc<G>(A[j], j);
      float inv = __frsqrt_rn(fmaxf(piv, MINVALF));
      // STRICTLY lower (the diagonal lives in d_dinv): the trailing update only needs L[k][j] of the rows k > j, and lane j's own
      // row is finished -- what the update does to its entries right of the diagonal is never read
      float lj = (gq > j) ? A[j] * inv : 0.f;
      Lrow[j] = lj;
      if (gq == j) d_dinv = inv;
#pragma unroll
      for (int k = j + 1; k < NVP; k++) A[k] -= lj * bc<G>(lj, k);
    }
    // leave L in the dense LDS tile: the backward substitution reads its columns (= rows of L') from there
    if (g < NVP) {
#pragma unroll
      for (int k4 = 0; k4 < NVP / 4; k4++)
        *reinterpret_cast<float4*>(W + o_tile + g * TD + 4 * k4) = make_float4(Lrow[4 * k4], Lrow[4 * k4 + 1], Lrow[4 * k4 + 2], Lrow[4 * k4 + 3]);
    } else d_dinv = 1.f;
    GSYNC();
    scale_rows();
  }
  // After a factorisation the row registers are rewritten for the substitution: Lrow[k] <- L[g][k] / L[g][g].  The factor leaves
  // STRICTLY lower rows (registers and LDS tile: zeros on and right of the diagonal, 1 / L[g][g] in d_dinv), so this is NVP plain
  // multiplies -- no `k < g` selects, whose 36 lane masks the compiler kept in SGPR pairs, spilled, and restored with two
  // v_readlane in every step of the substitution that follows.  With the lane's own 1 / L[g][g] folded into its row (and into its
  // right-hand side), step j of the forward substitution is "broadcast x_j, one fma" for every lane.
  __device__ __forceinline__ void scale_rows() {
#pragma unroll
    for (int k = 0; k < NVP; k++) Lrow[k] *= d_dinv;
  }

  // x <- (L L')^-1 x ; lane i holds x_i.  Lrow = the scaled rows (scale_rows), the LDS tile = the strictly lower part of L.
  __device__ __forceinline__ float solve(float x) const {
    // The column entries of the back substitution (L' z = y needs L[i][g], i > g: column g of the LDS tile, zero for i <= g) are
    // fetched FIRST, with unconditional loads pinned ahead of the forward chain: their LDS latency hides behind the forward
    // substitution.  (`g < i ? LT[..] : 0` compiled to a branch around a ds_read with a full s_waitcnt in EVERY step -- one LDS
    // round trip per step, 5.5 k of the 6.3 k cycles of a 36-wide solve; unconditional but unpinned loads still waited once per
    // two steps.)
    const float* LT = W + o_tile + (g < NVP ? g : 0);   // LT[i*TD] = L[i][g]
    const float ds = g < NVP ? d_dinv : 0.f;            // (lanes without a row read column 0: scaled to zero)
    float c[NVP];
#pragma unroll
    for (int i = 0; i < NVP; i++) c[i] = LT[i * TD];
    __builtin_amdgcn_sched_barrier(0);
    // L y = b:  x'_g = b_g / L_gg - sum_{k < g} (L_gk / L_gg) y_k, and y_j is lane j's x' once the steps k < j are in
    x *= d_dinv;
#pragma unroll
    for (int j = 0; j < NVP; j++) x = fmaf(-Lrow[j], bc<G>(x, j), x);
    // L' z = y:  x''_g = y_g / L_gg - sum_{k > g} (L_kg / L_gg) z_k; the column entries are scaled by the lane's own 1 / L_gg off
    // the dependent chain
    x *= d_dinv;
#pragma unroll
    for (int i = NVP - 1; i >= 0; i--) x = fmaf(-(c[i] * ds), bc<G>(x, i), x);
    return x;
  }

  // y_i = sum_j M[i][j] x_j
  __device__ __forceinline__ float mul_m(float x) const {
    float y = 0.f;
    if constexpr (SP) return sp_mul_m(x);
    if constexpr (LDS_VECTOR) {
      // x through LDS (one write, NVP/4 broadcast 128-bit reads) instead of NVP cross-lane broadcasts
      float* X = W + KL().xvec;
      if (g < NVP) X[g] = x;
#pragma unroll
      for (int k4 = 0; k4 < NVP / 4; k4++) {
        const float4 r = *reinterpret_cast<const float4*>(X + 4 * k4);
        y += Mrow[4 * k4] * r.x + Mrow[4 * k4 + 1] * r.y + Mrow[4 * k4 + 2] * r.z + Mrow[4 * k4 + 3] * r.w;
      }
      return y;
    }
#pragma unroll
    for (int j = 0; j < NVP; j++) y += Mrow[j] * bc<G>(x, j);
    return y;
  }

  // ------------------------------------------- A5/A6 passive + actuation -> qfrc_smooth
  __device__ __forceinline__ void passive_actuation() {
    if (!(TW && a.two_wave)) actuation();
    else tw_wait(1, tw_n);
    smooth_force();
  }
  // tendon springs / dampers, actuator dynamics and forces, J'f: LDS in, LDS out (a two-wave launch runs it in the helper wave)
  __device__ __forceinline__ void actuation() {
    const auto& L = KL();
    if constexpr (IMPL) {   // implicitfast: start the velocity-derivative weights from the passive dampers (mjd_passive_vel)
      for (int t = g; t < KD().ntendon; t += G) W[L.tenw + t] = MF_(TENDON_DAMPING)[t];
      if (g < KD().nv) W[L.dofw + g] = MF_(DOF_DAMPING)[g];
    }
    for (int t = g; t < KD().ntendon; t += G) {
      float k = MF_(TENDON_STIFFNESS)[t], bd = MF_(TENDON_DAMPING)[t], f = 0.f;
      if (k != 0.f || bd != 0.f) {
        float len = W[L.tenlen + t], lo = MF_(TENDON_LENGTHSPRING)[2 * t], hi = MF_(TENDON_LENGTHSPRING)[2 * t + 1];
        if (len > hi) f = k * (hi - len);
        else if (len < lo) f = k * (lo - len);
        f -= bd * W[L.tenvel + t];
      }
      W[L.tenfrc + t] = f;
    }
    if (g < KD().nv) W[L.vec + g] = 0.f;
    GSYNC();
    // section offsets of the actuator tables, pinned for the loop (see PIN_S)
    int s_cl = SECOFF_(ACT_CTRLLIMITED), s_cr = SECOFF_(ACT_CTRLRANGE), s_aa = SECOFF_(ACT_ACTADR), s_id = SECOFF_(ACT_TRNID),
        s_gr = SECOFF_(ACT_GEAR), s_tt = SECOFF_(ACT_TRNTYPE), s_dt = SECOFF_(ACT_DYNTYPE), s_dp = SECOFF_(ACT_DYNPRM),
        s_lr = SECOFF_(ACT_LENGTHRANGE), s_a0 = SECOFF_(ACT_ACC0), s_gt = SECOFF_(ACT_GAINTYPE), s_gp = SECOFF_(ACT_GAINPRM),
        s_bt = SECOFF_(ACT_BIASTYPE), s_bp = SECOFF_(ACT_BIASPRM), s_fl = SECOFF_(ACT_FORCELIMITED), s_fr = SECOFF_(ACT_FORCERANGE);
    PIN_S(s_cl); PIN_S(s_cr); PIN_S(s_aa); PIN_S(s_id); PIN_S(s_gr); PIN_S(s_tt); PIN_S(s_dt); PIN_S(s_dp); PIN_S(s_lr); PIN_S(s_a0);
    PIN_S(s_gt); PIN_S(s_gp); PIN_S(s_bt); PIN_S(s_bp); PIN_S(s_fl); PIN_S(s_fr);
    for (int u = g; u < KD().nu; u += G) {
      // Every table word of actuator u is requested up front, unconditionally: flag by flag (`if (limited[u]) .. range[u]`,
      // three looks at dyntype, gain / bias type, ...) the loop was a chain of a dozen load -> wait -> branch links per sweep.
      const int f_cl = AI_(s_cl)[u], aa = AI_(s_aa)[u], id = AI_(s_id)[u], f_tt = AI_(s_tt)[u], f_dt = AI_(s_dt)[u],
                f_gt = AI_(s_gt)[u], f_bt = AI_(s_bt)[u], f_fl = AI_(s_fl)[u];
      const float cr0 = AF_(s_cr)[2 * u], cr1 = AF_(s_cr)[2 * u + 1], gear = AF_(s_gr)[u];
      const float lr0 = AF_(s_lr)[2 * u], lr1 = AF_(s_lr)[2 * u + 1], acc0 = AF_(s_a0)[u];
      const float flo = AF_(s_fr)[2 * u], fhi = AF_(s_fr)[2 * u + 1];
      float dp[3], gp[9], bp[9];
#pragma unroll
      for (int k = 0; k < 3; k++) dp[k] = AF_(s_dp)[3 * u + k];
#pragma unroll
      for (int k = 0; k < 9; k++) { gp[k] = AF_(s_gp)[9 * u + k]; bp[k] = AF_(s_bp)[9 * u + k]; }
      float ctrl = W[L.ctrl + u];
      const float actv = W[L.act + (aa >= 0 ? aa : 0)];
      if (f_cl) ctrl = clampf(ctrl, cr0, cr1);
      float len, vel, input = ctrl;
      const bool ten = f_tt == MM_TRN_TENDON;
      if (ten) { len = gear * W[L.tenlen + id]; vel = gear * W[L.tenvel + id]; }
      else { len = gear * W[L.qpos + MI_(JNT_QPOSADR)[id]]; vel = gear * W[L.qvel + MI_(JNT_DOFADR)[id]]; }
      if (f_dt == MM_DYN_MUSCLE) {
        W[L.actdot + aa] = muscle_dynamics(ctrl, actv, dp);
        input = actv;
      } else if (f_dt == MM_DYN_INTEGRATOR) {
        W[L.actdot + aa] = ctrl; input = actv;
      } else if (f_dt == MM_DYN_FILTER) {
        W[L.actdot + aa] = (ctrl - actv) / fmaxf(MINVALF, dp[0]); input = actv;
      }
      float gain, bias = 0.f;
      if (f_gt == MM_GAIN_MUSCLE) gain = muscle_gain(len, vel, lr0, lr1, acc0, gp);
      else gain = gp[0];
      if (f_bt == MM_BIAS_MUSCLE) bias = muscle_bias(len, lr0, lr1, acc0, bp);
      else if (f_bt == MM_BIAS_AFFINE)   // position / velocity servos
        bias = bp[0] + bp[1] * len + bp[2] * vel;
      float f = gain * input + bias;
      bool clamped = false;
      if (f_fl) {
        f = clampf(f, flo, fhi);
        clamped = f <= flo || f >= fhi;
      }
      W[L.actfrc + u] = f; W[L.actlen + u] = len; W[L.actvel + u] = vel;
      if (ten) atomicAdd(&W[L.tenfrc + id], gear * f);
      else atomicAdd(&W[L.vec + MI_(JNT_DOFADR)[id]], gear * f);
      if constexpr (IMPL) {
        // s = d force / d velocity (mjd_actuator_vel: bias_vel + gain_vel * input; none while the force sits on its range)
        float s = 0.f;
        if (f_bt == MM_BIAS_AFFINE) s = bp[2];
        if (f_gt == MM_GAIN_MUSCLE) {
          const float* prm = gp;
          const float force = muscle_f0(prm, acc0);
          const float L0 = (lr1 - lr0) / fmaxf(MINVALF, prm[1] - prm[0]);
          const float Ln = prm[0] + (len - lr0) / fmaxf(MINVALF, L0);
          const float vs = fmaxf(MINVALF, L0 * prm[6]), V = vel / vs;
          const float fvmax = prm[8], y = fvmax - 1.f;
          const float dFV = V <= -1.f ? 0.f : (V <= 0.f ? 2.f * (V + 1.f) : (V <= y ? 2.f * (y - V) / fmaxf(MINVALF, y) : 0.f));
          s += -force * muscle_fl(Ln, prm[4], prm[5]) * dFV / vs * input;
        }
        if (clamped) s = 0.f;
        if (s != 0.f) {
          if (ten) atomicAdd(&W[L.tenw + id], -s * gear * gear);
          else atomicAdd(&W[L.dofw + MI_(JNT_DOFADR)[id]], -s * gear * gear);
        }
      }
    }
    GSYNC();
    // J' f: every tendon lane scatters its (<= 8) Jacobian entries into the per-dof accumulator with LDS float
    // atomics (one wave => deterministic lane order); shorter critical path than gathering ~25 entries per wrist dof
    {
      int s_ja = SECOFF_(TENJ_ADR), s_jd = SECOFF_(TENJ_DOF), o_tj = L.tenj, o_vec = L.vec, o_tf = L.tenfrc;
      PIN_S(s_ja); PIN_S(s_jd); PIN_S(o_tj); PIN_S(o_vec); PIN_S(o_tf);
      for (int t = g; t < KD().ntendon; t += G) {
        float f = W[o_tf + t];
        const int e0 = AI_(s_ja)[t], e1 = AI_(s_ja)[t + 1];
        if (f != 0.f)
          for (int e = e0; e < e1; e++) atomicAdd(&W[o_vec + AI_(s_jd)[e]], W[o_tj + e] * f);
      }
    }
    GSYNC();
  }
  // qfrc_smooth of dof g: passive joint forces - bias + actuation
  __device__ __forceinline__ void smooth_force() {
    const auto& L = KL();
    d_smooth = 0.f;
    if (g < KD().nv) {
      float s = -MF_(DOF_DAMPING)[g] * d_qvel - d_bias + W[L.vec + g];
      if constexpr (GEN) {
        // (general-row kernels: the joint's words at once, position and spring reference unconditionally -- two round trips, no
        // branch in between: leg +1 %.  The same form costs the 250-VGPR hand kernel 0.7 %: it keeps the chain.)
        const int j = c_rowj;                       // = DOF_JNTID[g], held since the constructor
        const float ks = MF_(JNT_STIFFNESS)[j];
        const int type = MI_(JNT_TYPE)[j], qa = MI_(JNT_QPOSADR)[j];
        const float qs = W[L.qpos + qa], q0s = MF_(QPOS_SPRING)[qa];
        if (ks != 0.f && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) s -= ks * (qs - q0s);
      } else {
        int j = MI_(DOF_JNTID)[g];
        float ks = MF_(JNT_STIFFNESS)[j];
        int type = MI_(JNT_TYPE)[j];
        if (ks != 0.f && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) {
          int qa = MI_(JNT_QPOSADR)[j];
          s -= ks * (W[L.qpos + qa] - MF_(QPOS_SPRING)[qa]);
        }
      }
      d_smooth = s;
    }
  }

  // ---------------------------------------------------------------- Newton solver (A7)
  // total cost of candidate x (dof lanes) given Ma = M x; also latches r_jar
  __device__ __forceinline__ float cost_of(float x, float Ma) {
    float c = 0.5f * (x - d_qaccsm) * (Ma - d_smooth);
    float xr = sh<G>(x, r_dof);
    r_jar = r_sign * xr - r_aref;
    if (r_active && r_jar < 0.f) c += 0.5f * r_D * r_jar * r_jar;
    return gsum<G>(c);
  }

  __device__ __forceinline__ void solve_constraints() {
    if constexpr (GEN) { solve_constraints_gen(); return; }
    const int nv = KD().nv;
    niter = 0;
    d_qfrccon = 0.f;
    if (nefc == 0) { d_qacc = d_qaccsm; return; }
    const float scale = 1.f / (KD().meaninertia * (float)(nv > 1 ? nv : 1));
#define PFN(stage, t0_) do { MM_FENCE(); if (MM_STAGE_PROF && a.prof) { const unsigned long long t1_ = clock64(); pf[MM_STAGE_PROF ? stage : 0] += t1_ - t0_; t0_ = t1_; } } while (0)
    unsigned long long tn_ = (MM_STAGE_PROF && a.prof) ? clock64() : 0;
    // warm start: qacc_warmstart is kept only if it beats the unconstrained solution
    float Ma_ws = mul_m(d_warm);
    float cost_ws = cost_of(d_warm, Ma_ws);
    float cost_sm = cost_of(d_qaccsm, d_smooth);
    float Ma;
    if (cost_ws < cost_sm) { d_qacc = d_warm; Ma = Ma_ws; (void)cost_of(d_qacc, Ma); }
    else { d_qacc = d_qaccsm; Ma = d_smooth; }
    // Termination in fp32: the cost (hundreds) carries ~1e-5 of rounding noise, far above MuJoCo's scaled
    // tolerance, so "improvement < tol" would stop with a residual gradient.  The cost is piecewise quadratic:
    // a FULL Newton step (alpha = 1) that leaves the active set unchanged lands on the exact minimiser, which is
    // the convergence test used here (the gradient test is kept for the exact-arithmetic case).
    PFN(PF_N_WARM, tn_);
    float alpha_prev = 0.f;
    unsigned long long set_prev = 0ull;
#if MM_NEWTON_POLISH
    bool polished = false;
#endif
    for (int iter = 0; iter < KD().iterations; iter++) {
      const bool on = r_active && r_jar < 0.f;
      const unsigned long long set_now = __ballot(on);
      float f = on ? -r_D * r_jar : 0.f;
      d_qfrccon = rows_to_dof(r_sign * f);
      float grad = g < nv ? Ma - d_smooth - d_qfrccon : 0.f;
      float gn = sqrtf(gsum<G>(grad * grad));
      if (scale * gn < KD().tolerance) break;
      if (iter > 0 && fabsf(alpha_prev - 1.f) < 1e-3f) {
        // compare the active sets of THIS group only
        const int lane = threadIdx.x & 63;
        const unsigned long long gm = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane - g);
        if (((set_now ^ set_prev) & gm) == 0ull) {
#if MM_NEWTON_POLISH
          if (polished) break;
          polished = true;     // experiment: one more Newton step from the (nominally exact) minimiser
#else
          break;
#endif
        }
      }
      set_prev = set_now;
      float dadd = rows_to_dof(on ? r_D : 0.f);
      PFN(PF_N_GRAD, tn_);
      float search = -factor_solve(dadd, grad);
      if (g >= nv) search = 0.f;
      PFN(PF_N_SOLVE, tn_);
      float sn = sqrtf(gsum<G>(search * search));
      if (sn < MINVALF) break;
      // (M + diag(dadd)) search = -grad, so M search needs no product: the residual of the solve is of the order of the
      // rounding of an explicit product
      float Mv;
      if constexpr (SP && !MM_NEWTON_TRUE_MV) Mv = g < nv ? -grad - dadd * search : 0.f;
      else Mv = mul_m(search);
      float jv = r_sign * sh<G>(search, r_dof);
      float dm = Ma - d_smooth;
      float q1 = gsum<G>(search * dm), q2 = gsum<G>(0.5f * search * Mv);
      const float gtol = KD().tolerance * KD().ls_tolerance * sn / scale;
      // exact line search on the convex piecewise-quadratic phi(alpha): safeguarded Newton on phi'(alpha) = 0
      float alpha = 1.f, lo = 0.f, hi = -1.f;
      for (int it = 0; it < KD().ls_iterations; it++) {
        float x = r_jar + alpha * jv;
        float d1 = 0.f, d2 = 0.f;
        if (r_active && x < 0.f) { d1 = r_D * x * jv; d2 = r_D * jv * jv; }
        d1 = gsum<G>(d1) + q1 + 2.f * alpha * q2;
        d2 = gsum<G>(d2) + 2.f * q2;
        if (fabsf(d1) < fmaxf(gtol, 1e-6f * fabsf(q1))) break;
        if (d1 < 0.f) lo = alpha; else hi = alpha;
        float next = alpha - d1 / fmaxf(d2, MINVALF);
        if (hi >= 0.f && (next <= lo || next >= hi)) next = 0.5f * (lo + hi);
        else if (hi < 0.f && next <= lo) next = 2.f * lo + 1e-10f;
        if (MM_LS_RELSTOP ? fabsf(next - alpha) <= MM_LS_RELSTOP_TOL * fabsf(alpha) : next == alpha) { alpha = next; break; }
        alpha = next;
      }
      if (!(alpha > 0.f)) break;
      d_qacc += alpha * search; Ma += alpha * Mv; r_jar += alpha * jv;
      alpha_prev = alpha;
      niter = iter + 1;
      // a step below fp32 resolution of qacc cannot improve the solution (rows sitting at jar ~ 0 would
      // otherwise toggle in and out of the active set for ever)
      {
        float stepmax = gmax<G>(fabsf(alpha * search)), qmax = gmax<G>(fabsf(d_qacc));
        if (stepmax <= 2e-7f * fmaxf(qmax, 1.f)) {
          const bool on2 = r_active && r_jar < 0.f;
          d_qfrccon = rows_to_dof(r_sign * (on2 ? -r_D * r_jar : 0.f));
          break;
        }
      }
      if (iter == KD().iterations - 1) {
        const bool on2 = r_active && r_jar < 0.f;
        d_qfrccon = rows_to_dof(r_sign * (on2 ? -r_D * r_jar : 0.f));
        status |= 4;
      }
      PFN(PF_N_LS, tn_);
    }
    PFN(PF_N_GRAD, tn_);
#undef PFN
  }


  // ===================================================== general constraint rows (GEN models)
  // Row r of efc_J lives in LDS with a 16-byte aligned row stride (NVP+4 words): the Hessian build and J x read rows with
  // 128-bit LDS loads (a wave-uniform row is one broadcast ds_read_b128 per four columns); lane r owns the row's scalars
  // (D, aref, jar).  Row order: equalities, active joint
  // limits (compacted), contact pyramid edges (compacted).  Restates mmo_make_constraint / mmo_collision.inc.
  static constexpr int RS = NVP + 4;
  __device__ __forceinline__ float* Jrow(int r) const { return W + KL().efcJ + (r < KD().efc_rows ? r : 0) * RS; }
  // exclusive prefix count of a 0/1 flag over the lanes of the group: ballot + population count of the lower lanes (three
  // instructions; a shuffle scan is log2(G) dependent trips through the LDS crossbar, ~800 cycles for G = 64)
  __device__ __forceinline__ int gscan_flag(bool f) const {
    const unsigned long long m = __ballot(f);
    const int lane = threadIdx.x & 63;
    unsigned long long below = m & ((1ull << lane) - 1ull);
    if constexpr (G < 64) below &= ((1ull << G) - 1ull) << (lane - g);      // this group's lanes only
    return __popcll(below);
  }
  // exclusive prefix sum of small non-negative counts (<= 15): one flag scan per bit
  __device__ __forceinline__ int gscan_excl(int v) const {
    return gscan_flag(v & 1) + 2 * gscan_flag(v & 2) + 4 * gscan_flag(v & 4) + 8 * gscan_flag(v & 8);
  }
  __device__ __forceinline__ int gsum_i(int v) const { return (int)(gsum<G>((float)v) + 0.5f); }   // counts <= 64: exact
  __device__ __forceinline__ V3 geom_zaxis(int gi) const {
    M3 R = geom_mat(gi);
    return v3(R.m[2], R.m[5], R.m[8]);
  }
  // sphere-sphere building block (mmo_collision.inc: sphere_sphere); returns false when dist >= margin
  __device__ __forceinline__ bool sph_sph(V3 c1, float r1, V3 c2, float r2, float margin, float& dist, V3& pos, V3& n) const {
    V3 d = c2 - c1;
    float len = sqrtf(dot(d, d));
    dist = len - r1 - r2;
    if (!(dist < margin)) return false;
    n = len < MINVALF ? v3(0.f, 0.f, 1.f) : (1.f / len) * d;
    pos = c1 + (r1 + 0.5f * dist) * n;
    return true;
  }
  __device__ __forceinline__ bool pln_sph(V3 pp, V3 pn, V3 c, float r, float margin, float& dist, V3& pos, V3& n) const {
    dist = dot(c - pp, pn) - r;
    if (!(dist < margin)) return false;
    n = pn;
    pos = c - (r + 0.5f * dist) * pn;
    return true;
  }
  __device__ __forceinline__ V3 seg_closest(V3 a0, V3 u, float h, V3 p) const {
    float t = clampf(dot(p - a0, u), -h, h);
    return a0 + t * u;
  }

  __device__ __forceinline__ void make_constraint_gen() {
    const auto& L = KL();
    float* RT = W + L.rowtab;
    {
      float4* Jz = reinterpret_cast<float4*>(W + L.efcJ);
      for (int e = g; e < KD().efc_rows * RS / 4; e += G) Jz[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    GSYNC();
    const int neq = KD().neq;
    // ---- equality rows: joint coupling q1 - q1_0 = poly(q2 - q2_0)
    if (g < neq) {
      const int e = g, j1 = MI_(EQ_OBJ1ID)[e], j2 = MI_(EQ_OBJ2ID)[e];
      const float* c = MF_(EQ_DATA) + 5 * e;
      const int q1 = MI_(JNT_QPOSADR)[j1], d1 = MI_(JNT_DOFADR)[j1];
      float pos1 = W[L.qpos + q1] - MF_(QPOS0)[q1], res, deriv = 0.f;
      float dA = MF_(DOF_INVWEIGHT0)[d1];
      if (j2 >= 0) {
        const int q2 = MI_(JNT_QPOSADR)[j2], d2 = MI_(JNT_DOFADR)[j2];
        float x = W[L.qpos + q2] - MF_(QPOS0)[q2];
        res = pos1 - (c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * c[4]))));
        deriv = c[1] + x * (2.f * c[2] + x * (3.f * c[3] + x * 4.f * c[4]));
        dA += MF_(DOF_INVWEIGHT0)[d2];
        Jrow(e)[d2] = -deriv;
      } else res = pos1 - c[0];
      Jrow(e)[d1] = 1.f;
      RT[3 * e] = __int_as_float(MM_CON_EQUALITY | (e << 3)); RT[3 * e + 1] = res; RT[3 * e + 2] = dA;
    }
    // ---- dof friction loss (MuJoCo row order: equality, friction loss, limits, contacts): J = e_dof at pos 0
    int nfr = 0;
    if (KD().nfric) {
      const int fr = (g < KD().nv && MF_(DOF_FRICTIONLOSS)[g] > 0.f) ? 1 : 0;
      const int frank = gscan_flag(fr != 0);
      nfr = gsum_i(fr);
      if (fr) {
        const int r = neq + frank;
        if (r < KD().efc_rows) {
          Jrow(r)[g] = 1.f;
          RT[3 * r] = __int_as_float(MM_CON_FRICTION_DOF | (g << 3)); RT[3 * r + 1] = 0.f; RT[3 * r + 2] = MF_(DOF_INVWEIGHT0)[g];
        }
      }
    }
    // ---- joint limits, compacted behind the equalities
    int lim = 0, ldof = 0;
    float ldist = 0.f, lsign = 1.f, lmargin = 0.f;
    if (g < KD().njnt) {
      // (all table words of the joint at once, then its position: no load -> branch -> load chain)
      const int j = g, type = MI_(JNT_TYPE)[j], limited = MI_(JNT_LIMITED)[j], da = MI_(JNT_DOFADR)[j], qa = MI_(JNT_QPOSADR)[j];
      const float mg = MF_(JNT_MARGIN)[j], rlo = MF_(JNT_RANGE)[2 * j], rhi = MF_(JNT_RANGE)[2 * j + 1];
      const float q = W[L.qpos + qa];
      if (limited && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) {
        ldof = da;
        lmargin = mg;
        float dlo = q - rlo, dhi = rhi - q;
        ldist = dlo;
        if (!(dlo < lmargin) && dhi < lmargin) { ldist = dhi; lsign = -1.f; }
        lim = ldist < lmargin ? 1 : 0;
      }
    }
    const int lrank = gscan_flag(lim != 0), nlim = gsum_i(lim);
    int over = 0;
    if (lim) {
      const int r = neq + nfr + lrank;
      if (r < KD().efc_rows) {
        Jrow(r)[ldof] = lsign;
        RT[3 * r] = __int_as_float(MM_CON_LIMIT_JOINT | (g << 3)); RT[3 * r + 1] = ldist - lmargin; RT[3 * r + 2] = MF_(DOF_INVWEIGHT0)[ldof];
      } else over = 1;
    }
    // ---- tendon limits, compacted behind the joint limits (MuJoCo row order); J = -+ the tendon's Jacobian row
    int ntl = 0;
    if (KD().ntlim) {
      for (int t0 = 0; t0 < KD().ntendon; t0 += G) {
        const int t = t0 + g;
        int tl = 0;
        float tdist = 0.f, tsign = 1.f, tmargin = 0.f;
        if (t < KD().ntendon && MI_(TENDON_LIMITED)[t]) {
          const float len = W[L.tenlen + t];
          tmargin = MF_(TENDON_MARGIN)[t];
          const float dlo = len - MF_(TENDON_RANGE)[2 * t], dhi = MF_(TENDON_RANGE)[2 * t + 1] - len;
          tdist = dlo;
          if (!(dlo < tmargin) && dhi < tmargin) { tdist = dhi; tsign = -1.f; }
          tl = tdist < tmargin ? 1 : 0;
        }
        const int trank = gscan_flag(tl != 0), tcnt = gsum_i(tl);
        if (tl) {
          const int r = neq + nfr + nlim + ntl + trank;
          if (r < KD().efc_rows) {
            for (int e = MI_(TENJ_ADR)[t]; e < MI_(TENJ_ADR)[t + 1]; e++) Jrow(r)[MI_(TENJ_DOF)[e]] = tsign * W[L.tenj + e];
            RT[3 * r] = __int_as_float(MM_CON_LIMIT_TENDON | (t << 3)); RT[3 * r + 1] = tdist - tmargin; RT[3 * r + 2] = MF_(TENDON_INVWEIGHT0)[t];
          } else over = 1;
        }
        ntl += tcnt;
      }
    }
    // ---- contacts: lane p handles explicit pair p (up to two contacts for plane-capsule)
    const unsigned long long tc0_ = (MM_STAGE_PROF && a.prof) ? clock64() : 0;
    int nc = 0, rowsper = 0, b1 = 0, b2 = 0;
    float cdist[2] = {0.f, 0.f}, mu = 0.f, incl = 0.f;
    V3 cpos[2], cn[2];
    cpos[0] = cpos[1] = cn[0] = cn[1] = v3(0.f, 0.f, 0.f);
    if (g < KD().npair) {
      const int p = g, g1 = MI_(PAIR_GEOM1)[p], g2 = MI_(PAIR_GEOM2)[p];
      int t1 = MI_(GEOM_TYPE)[g1], t2 = MI_(GEOM_TYPE)[g2];
      if (env_gtype >= 0) {   // per-env model delta: type of one geom (mm_state.geom_type_env)
        if (g1 == a.s.geom_env_id) t1 = env_gtype;
        if (g2 == a.s.geom_env_id) t2 = env_gtype;
      }
      const float margin = MF_(PAIR_MARGIN)[p];
      incl = margin - MF_(PAIR_GAP)[p];
      mu = MF_(PAIR_FRICTION)[3 * p];
      rowsper = MI_(PAIR_CONDIM)[p] == 1 ? 1 : 4;
      b1 = MI_(GEOM_BODYID)[g1]; b2 = MI_(GEOM_BODYID)[g2];
      V3 x1 = geom_pos(g1), x2 = geom_pos(g2);
      float r1 = MF_(GEOM_SIZE)[3 * g1], h1 = MF_(GEOM_SIZE)[3 * g1 + 1];
      float r2 = MF_(GEOM_SIZE)[3 * g2], h2 = MF_(GEOM_SIZE)[3 * g2 + 1];
      if (env_has_gs) {   // per-env model delta: size of one geom (mm_state.geom_size_env)
        if (g1 == a.s.geom_env_id) { r1 = env_gsv[0]; h1 = env_gsv[1]; }
        if (g2 == a.s.geom_env_id) { r2 = env_gsv[0]; h2 = env_gsv[1]; }
      }
      if (t1 == MM_GEOM_PLANE && t2 == MM_GEOM_SPHERE) {
        nc = pln_sph(x1, geom_zaxis(g1), x2, r2, margin, cdist[0], cpos[0], cn[0]) ? 1 : 0;
      } else if (t1 == MM_GEOM_PLANE && t2 == MM_GEOM_CAPSULE) {
        V3 pn = geom_zaxis(g1), u2 = geom_zaxis(g2);
        for (int sgn = 0; sgn < 2; sgn++) {
          V3 c = x2 + (sgn ? h2 : -h2) * u2;
          float dd; V3 pp, nn;
          if (pln_sph(x1, pn, c, r2, margin, dd, pp, nn)) { cdist[nc] = dd; cpos[nc] = pp; cn[nc] = nn; nc++; }
        }
      } else if (t1 == MM_GEOM_SPHERE && t2 == MM_GEOM_SPHERE) {
        nc = sph_sph(x1, r1, x2, r2, margin, cdist[0], cpos[0], cn[0]) ? 1 : 0;
      } else if (t1 == MM_GEOM_SPHERE && t2 == MM_GEOM_CAPSULE) {
        V3 c = seg_closest(x2, geom_zaxis(g2), h2, x1);
        nc = sph_sph(x1, r1, c, r2, margin, cdist[0], cpos[0], cn[0]) ? 1 : 0;
      } else if (t1 == MM_GEOM_CAPSULE && t2 == MM_GEOM_CAPSULE) {
        V3 u1 = geom_zaxis(g1), u2 = geom_zaxis(g2), w = x1 - x2;
        float bb = dot(u1, u2), dd = dot(u1, w), ee = dot(u2, w), den = 1.f - bb * bb;
        float s1 = den < 1e-9f ? 0.f : (bb * ee - dd) / den;
        s1 = clampf(s1, -h1, h1);
        float s2 = ee + bb * s1;
        if (s2 < -h2 || s2 > h2) { s2 = s2 < -h2 ? -h2 : h2; s1 = clampf(bb * s2 - dd, -h1, h1); }
        nc = sph_sph(x1 + s1 * u1, r1, x2 + s2 * u2, r2, margin, cdist[0], cpos[0], cn[0]) ? 1 : 0;
      } else if ((t1 == MM_GEOM_CAPSULE && t2 >= MM_GEOM_ELLIPSOID) || (t2 == MM_GEOM_CAPSULE && t1 >= MM_GEOM_ELLIPSOID)) {
        // capsule vs ellipsoid / cylinder / box (mmo_collision.inc: capsule_convex); flip: the convex geom is geom1
        const bool flip = t2 == MM_GEOM_CAPSULE;
        const int gc = flip ? g2 : g1, gs = flip ? g1 : g2, ts = flip ? t1 : t2;
        const V3 xc = flip ? x2 : x1, xs = flip ? x1 : x2;
        const float rc = flip ? r2 : r1, hc = flip ? h2 : h1;
        V3 ss = ld3(MF_(GEOM_SIZE) + 3 * gs);
        if (env_has_gs && gs == a.s.geom_env_id) ss = v3(env_gsv[0], env_gsv[1], env_gsv[2]);
        const V3 uc = geom_zaxis(gc);
        const M3 ms = geom_mat(gs);
        const SegHit hit = seg_shape_call(ts, ss, mtv(ms, xc - xs), mtv(ms, uc), hc);
        const float tt = hit.t, sd = hit.sd;
        const V3 gsh = hit.g;
        const float dd = sd - rc;
        if (dd < margin) {
          V3 gw = mv(ms, gsh);
          cdist[0] = dd;
          cpos[0] = (xc + tt * uc) - (rc + 0.5f * dd) * gw;
          cn[0] = flip ? gw : -1.f * gw;
          nc = 1;
        }
      }
    }
    if (MM_STAGE_PROF && a.prof) pf[MM_STAGE_PROF ? PF_IO : 0] += clock64() - tc0_;   // narrow phase only (reported as 'io' = collide)
    int myrows = 0;
    for (int c = 0; c < 2; c++) if (c < nc && cdist[c] < incl) myrows += rowsper;
    int base = neq + nfr + nlim + ntl + gscan_excl(myrows);
    // Row table entries by the pair's lane; the Jacobian rows by ALL lanes of the group, one dof each: lane i holds its dof's
    // motion axis in registers and knows from a per-body chain mask (Aux.body_dofmask) whether dof i moves geom1's or geom2's
    // body, so a contact costs a handful of broadcasts + 4 stores per lane instead of one lane walking two kinematic chains
    // with two dependent LDS round trips per dof (11 k cycles per forward pass for the leg's foot contacts).
    int cbase[2] = {-1, -1};
    int made = 0;      // contact rows this lane actually created (a contact that does not fit is dropped whole)
    for (int c = 0; c < 2; c++) {
      if (!(c < nc && cdist[c] < incl)) continue;
      if (base + rowsper > KD().efc_rows) { over = 1; continue; }
      cbase[c] = base;
      made += rowsper;
      const float tran = MF_(BODY_INVWEIGHT0)[2 * b1] + MF_(BODY_INVWEIGHT0)[2 * b2];
      for (int k = 0; k < rowsper; k++) {
        RT[3 * (base + k)] = __int_as_float(MM_CON_CONTACT | (g << 3));
        RT[3 * (base + k) + 1] = cdist[c] - incl;
        RT[3 * (base + k) + 2] = rowsper == 1 ? tran : tran + mu * mu * tran;
      }
      base += rowsper;
    }
    {
      const int np_ = KD().npair, nv_ = KD().nv;
      V3 off_c = v3(0.f, 0.f, 0.f);     // subtree COM of this lane's dof (cdof is expressed about it)
      if (g < nv_) off_c = ld3(W + L.com + 3 * AUXI(dof_rootslot)[g]);
      const int* bmask = AUXI(body_dofmask);
      // pairs that made a contact in any env group of this wave (lane p = pair p): only those are visited -- the scan over all
      // pairs cost a broadcast + ballot + branch per (pair, contact) slot, 40 of them per pass for the hand's 20 pairs
      unsigned long long act = __ballot(cbase[0] >= 0 || cbase[1] >= 0);
      if constexpr (G < 64) {
#pragma unroll
        for (int sft = G; sft < 64; sft <<= 1) act |= act >> sft;
        act &= (1ull << G) - 1ull;
      }
      (void)np_;
      while (act) {
        const int p = __builtin_ctzll(act);
        act &= act - 1ull;
        for (int c = 0; c < 2; c++) {
          const int rb = (int)bc<G>((float)cbase[c], p);                  // row base of contact c of pair p in THIS group (-1: none)
          if (__ballot(rb >= 0) == 0ull) continue;
          const int pb1 = (int)bc<G>((float)b1, p), pb2 = (int)bc<G>((float)b2, p), prow = (int)bc<G>((float)rowsper, p);
          // the two bodies' dof masks: one unconditional word load each, both in flight (`g < 32 ? lo : hi` was two branches
          // with a load and a full wait apiece, ahead of every contact's Jacobian)
          const int w1 = bmask[2 * pb1 + ((g >> 5) & 1)], w2 = bmask[2 * pb2 + ((g >> 5) & 1)];
          const V3 pn = v3(bc<G>(cn[c].x, p), bc<G>(cn[c].y, p), bc<G>(cn[c].z, p));
          const V3 pp = v3(bc<G>(cpos[c].x, p), bc<G>(cpos[c].y, p), bc<G>(cpos[c].z, p));
          const float pmu = bc<G>(mu, p);
          if (rb < 0 || g >= nv_) continue;
          const bool in1 = (w1 >> (g & 31)) & 1, in2 = (w2 >> (g & 31)) & 1;
          if (in1 == in2) continue;                                       // on neither chain, or on both (the two terms cancel)
          // contact frame (mmo_collision.inc: make_frame)
          V3 y = (pn.y < 0.5f && pn.y > -0.5f) ? v3(0.f, 1.f, 0.f) : v3(0.f, 0.f, 1.f);
          y = y - dot(pn, y) * pn;
          y = (1.f / fmaxf(sqrtf(dot(y, y)), MINVALF)) * y;
          const V3 z = cross(pn, y);
          const V3 ang = v3(d_cdof[0], d_cdof[1], d_cdof[2]), lin = v3(d_cdof[3], d_cdof[4], d_cdof[5]);
          const V3 v = lin + cross(ang, pp - off_c);
          const float sg = in2 ? 1.f : -1.f;
          const float vn = sg * dot(pn, v), v1 = sg * pmu * dot(y, v), v2 = sg * pmu * dot(z, v);
          if (prow == 1) Jrow(rb)[g] = vn;
          else { Jrow(rb)[g] = vn + v1; Jrow(rb + 1)[g] = vn - v1; Jrow(rb + 2)[g] = vn + v2; Jrow(rb + 3)[g] = vn - v2; }
        }
      }
    }
    if (gor<G>(over)) status |= 8;   // more rows than lanes: surplus rows dropped (njmax-style warning)
    // rows that exist: everything ahead of the contacts up to the table size, plus the contact rows that were created.  (Counting
    // the rows of a DROPPED contact -- round 2: min(sum, efc_rows) -- left the tail rows active with whatever the row table
    // held: a garbage row descriptor indexes the solimp / solref tables out of bounds, a memory fault with the model in HBM.)
    {
      const int pre = neq + nfr + nlim + ntl;
      nefc = (pre < KD().efc_rows ? pre : KD().efc_rows) + gsum_i(made);
    }
    {
      int w = nefc;
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) w = max(w, __shfl_xor(w, m, 64));
      nrows_wave = __builtin_amdgcn_readfirstlane(w);
    }
    GSYNC();
    // ---- owner stage: impedance / reference acceleration of row g (mmo_reference_constraint + pyramid R)
    r_active = g < nefc; r_eq = false; r_D = 0.f; r_aref = 0.f; r_jar = 0.f; r_floss = 0.f;
    if (r_active) {
      const int desc = __float_as_int(RT[3 * g]), kind = desc & 7, id = desc >> 3;
      const float x = RT[3 * g + 1], dA = RT[3 * g + 2];
      float vel = 0.f;
      {   // J_row . qvel (rows are 16-byte aligned: 128-bit loads; columns >= nv of a row are zero)
        // Straight-line over the padded width, every load unconditional and in flight at once; the tail words behind qvel[nv - 1]
        // are finite table words (act, ctrl) under zero columns of J, masked anyway.  (The trip count nv / 4 and the tail
        // predicates made this nine blocks with an LDS round trip each.)
        const float4* J4 = reinterpret_cast<const float4*>(Jrow(g));
        const int nv_ = KD().nv;
#pragma unroll
        for (int k4 = 0; k4 < NVP / 4; k4++) {
          const float4 j4 = J4[k4];
          const float* qv = W + L.qvel + 4 * k4;      // (the qvel table itself is only word-aligned)
          const float q0 = qv[0], q1 = qv[1], q2 = qv[2], q3 = qv[3];
          vel += (4 * k4 < nv_ ? j4.x * q0 : 0.f) + (4 * k4 + 1 < nv_ ? j4.y * q1 : 0.f) + (4 * k4 + 2 < nv_ ? j4.z * q2 : 0.f) +
                 (4 * k4 + 3 < nv_ ? j4.w * q3 : 0.f);
        }
      }
      const float *si, *sr;
      if (kind == MM_CON_EQUALITY) { si = MF_(EQ_SOLIMP) + 5 * id; sr = MF_(EQ_SOLREF) + 2 * id; }
      else if (kind == MM_CON_LIMIT_JOINT) { si = MF_(JNT_SOLIMP) + 5 * id; sr = MF_(JNT_SOLREF) + 2 * id; }
      else if (kind == MM_CON_LIMIT_TENDON) { si = MF_(TENDON_SOLIMP) + 5 * id; sr = MF_(TENDON_SOLREF) + 2 * id; }
      else if (kind == MM_CON_FRICTION_DOF) { si = MF_(DOF_SOLIMP) + 5 * id; sr = MF_(DOF_SOLREF) + 2 * id; r_floss = MF_(DOF_FRICTIONLOSS)[id]; }
      else { si = MF_(PAIR_SOLIMP) + 5 * id; sr = MF_(PAIR_SOLREF) + 2 * id; }
      impedance(si, sr, x, dA, vel, r_D, r_aref);
      if (kind == MM_CON_CONTACT && MI_(PAIR_CONDIM)[id] > 1) {
        const float m_ = MF_(PAIR_FRICTION)[3 * id];
        r_D = 1.f / fmaxf(MINVALF, 2.f * m_ * m_ / r_D);
      }
      r_eq = kind == MM_CON_EQUALITY;
    }
  }

  // (J x)_r for the row owned by this lane; x lives in the dof lanes
  __device__ __forceinline__ float jac_mul(float x) const {
    const float4* J = reinterpret_cast<const float4*>(Jrow(g));
    float s = 0.f;
    if constexpr (LDS_VECTOR) {
      // x through LDS (one write, NVP/4 broadcast 128-bit reads) instead of NVP cross-lane broadcasts
      float* X = W + KL().xvec;
      if (g < NVP) X[g] = x;
#pragma unroll
      for (int k = 0; k < NVP / 4; k++) {
        const float4 j4 = J[k], x4 = *reinterpret_cast<const float4*>(X + 4 * k);
        s += j4.x * x4.x + j4.y * x4.y + j4.z * x4.z + j4.w * x4.w;
      }
      return s;
    }
#pragma unroll
    for (int k = 0; k < NVP / 4; k++) {
      const float4 j4 = J[k];
      s += j4.x * bc<G>(x, 4 * k) + j4.y * bc<G>(x, 4 * k + 1) + j4.z * bc<G>(x, 4 * k + 2) + j4.w * bc<G>(x, 4 * k + 3);
    }
    return s;
  }
  // (J' f)_i for the dof owned by this lane; f lives in the row lanes
  __device__ __forceinline__ float jacT_mul(float f) const {
    const float* Jc = W + KL().efcJ + (g < NVP ? g : 0);
    float s = 0.f;
    // four rows per turn: the column loads are independent of the running sum (one row per turn exposes an LDS round trip per
    // row to a lone wave).  Rows past nefc hold zeros and carry no force; the table has a multiple of four rows.
    // ... and the next four are requested before the current four are used (the last turn re-reads its own rows)
    float j0 = Jc[0], j1 = Jc[RS], j2 = Jc[2 * RS], j3 = Jc[3 * RS];
    for (int r = 0; r < nrows_wave; r += 4) {
      const int rn = r + 4 < nrows_wave ? r + 4 : r;
      const float n0 = Jc[rn * RS], n1 = Jc[(rn + 1) * RS], n2 = Jc[(rn + 2) * RS], n3 = Jc[(rn + 3) * RS];
      s += j0 * bc<G>(f, r) + j1 * bc<G>(f, r + 1) + j2 * bc<G>(f, r + 2) + j3 * bc<G>(f, r + 3);
      j0 = n0; j1 = n1; j2 = n2; j3 = n3;
    }
    return g < KD().nv ? s : 0.f;
  }
  // force -s'(x) of the row owned by this lane at x = J a - aref; quad = the row is in its quadratic state (contributes
  // D J'J to the Hessian).  Equality rows are quadratic everywhere, limit / contact rows for x < 0, friction-loss rows are
  // Huber: the force -D x saturates at +-frictionloss (mmo_engine.c: row_cost)
  __device__ __forceinline__ float row_force(float x, bool& quad) const {
    // branch-free on purpose: the callers feed the result straight into wave-collective reductions
    const float f = -r_D * x;
    const bool fr = r_floss > 0.f;
    quad = r_active && (fr ? fabsf(f) < r_floss : (r_eq || x < 0.f));
    const float v = fr ? clampf(f, -r_floss, r_floss) : (quad ? f : 0.f);
    return r_active ? v : 0.f;
  }
  __device__ __forceinline__ float cost_gen(float x, float Ma) {
    float c = 0.5f * (x - d_qaccsm) * (Ma - d_smooth);
    r_jar = jac_mul(x) - r_aref;
    bool quad;
    const float f = row_force(r_jar, quad);
    if (quad) c += 0.5f * r_D * r_jar * r_jar;
    else if (r_floss > 0.f && r_active) c += r_floss * (fabsf(r_jar) - 0.5f * r_floss / r_D);
    (void)f;
    return gsum<G>(c);
  }

  __device__ __forceinline__ void solve_constraints_gen() {
    const int nv = KD().nv;
    niter = 0;
    d_qfrccon = 0.f;
    if (nrows_wave == 0) { d_qacc = d_qaccsm; return; }
    const float scale = 1.f / (KD().meaninertia * (float)(nv > 1 ? nv : 1));
#define PFN(stage, t0_) do { MM_FENCE(); if (MM_STAGE_PROF && a.prof) { const unsigned long long t1_ = clock64(); pf[MM_STAGE_PROF ? stage : 0] += t1_ - t0_; t0_ = t1_; } } while (0)
    unsigned long long tn_ = (MM_STAGE_PROF && a.prof) ? clock64() : 0;
    float Ma_ws = mul_m(d_warm);
    float cost_ws = cost_gen(d_warm, Ma_ws);
    float cost_sm = cost_gen(d_qaccsm, d_smooth);
    float Ma;
    if (cost_ws < cost_sm) { d_qacc = d_warm; Ma = Ma_ws; (void)cost_gen(d_qacc, Ma); }
    else { d_qacc = d_qaccsm; Ma = d_smooth; }
    PFN(PF_N_WARM, tn_);
    float alpha_prev = 0.f;
    unsigned long long set_prev = 0ull, sat_prev = 0ull;
    bool done = nefc == 0;     // envs of the wave that have no rows idle through the loop (wave-collective code below)
    for (int iter = 0; iter < KD().iterations; iter++) {
      bool on;
      const float rf = row_force(r_jar, on);
      // active-set signature: quadratic rows, plus the sign of saturated friction rows
      const unsigned long long set_now = __ballot(on), sat_now = KD().nfric ? __ballot(rf > 0.f && !on) : 0ull;
      d_qfrccon = jacT_mul(rf);
      float grad = g < nv ? Ma - d_smooth - d_qfrccon : 0.f;
      float gn = sqrtf(gsum<G>(grad * grad));
      if (scale * gn < KD().tolerance) done = true;
      if (!done && iter > 0 && fabsf(alpha_prev - 1.f) < 1e-3f) {
        const int lane = threadIdx.x & 63;
        const unsigned long long gm = (G == 64 ? ~0ull : ((1ull << G) - 1ull)) << (lane - g);
        if ((((set_now ^ set_prev) | (sat_now ^ sat_prev)) & gm) == 0ull) done = true;
      }
      if (__ballot(!done) == 0ull) break;
      set_prev = set_now; sat_prev = sat_now;
      PFN(PF_N_GRAD, tn_);
      // H = M + J_A' D J_A
      float A[NVP];
#if MM_MFMA_HBUILD
      if constexpr (G == 64) {
        // One env per wave: the rank-nefc update J' D J is a (NVP x K)(K x NVP) product -- the one GEMM-shaped piece of the step
        // -- and goes through the matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, same peak rate as the vector
        // pipe, but 64 multiply-adds per lane-instruction instead of 1) on 16 x 16 output tiles, upper triangle only.  Operands
        // come straight from the row-major efc_J table in LDS (A[i][k] = J[k][i], B[k][j] = D_k J[k][j]: the same load serves
        // both), the tiles go back through the dense LDS tile into the row-per-lane layout the Cholesky wants.  Replaces a loop
        // over the active rows (broadcast D_r, nine 128-bit row loads, NVP FMAs per row: ~550 cycles x ~30 rows per iteration).
        constexpr int NT = (NVP + 15) / 16;
        typedef float f4v __attribute__((ext_vector_type(4)));
        float* Dv = W + KL().rowtab;              // row table of make_constraint: dead since the owner stage
        Dv[g] = on ? r_D : 0.f;
        GSYNC();
        const int lr = g & 15, lk = g >> 4;
        const float* Jb = W + KL().efcJ;
        const int erows = KD().efc_rows;
        const int K4 = (nrows_wave + 3) >> 2;
        float* T = W + KL().u1;                   // the dense tile: the previous factor in there is dead
        // one tile ROW (ti) at a time: NT - ti accumulator tiles live instead of NT (NT + 1) / 2 (12 instead of 24 VGPRs for the
        // 36-dof leg, whose kernel sits at the 256-VGPR limit); the K sweep is repeated per tile row, its operand loads are cheap
#ifndef MM_HBUILD_ONE_SWEEP
#define MM_HBUILD_ONE_SWEEP 1
#endif
        if constexpr (MM_HBUILD_ONE_SWEEP != 0) {
          // ONE K sweep with all NT (NT + 1) / 2 accumulator tiles.  Per K step the sweeps by tile row issued NT, NT - 1, ... matrix
          // instructions behind one LDS round trip each -- all but the first are latency-bound (64 / 32 cycles of MFMA against a
          // ~100-cycle round trip for a lone wave); one sweep issues the same instructions behind a single round trip.  (24
          // accumulator VGPRs for the 48-wide tile of the 36-dof leg instead of 12: no new spills, leg +3.8 %.)
          f4v acc[NT][NT];
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = 0; tj < NT; tj++) acc[ti][tj] = f4v{0.f, 0.f, 0.f, 0.f};
          auto fetch = [&](const int kb, float& dsc, float (&av)[NT]) __attribute__((always_inline)) {
            const int row = 4 * kb + lk;
            const bool rok = row < erows;
            const int rr = rok ? row : 0;
            const float* Jr = Jb + rr * RS;
            const float dl = Dv[rr];
#pragma unroll
            for (int t = 0; t < NT; t++) {
              const int col = 16 * t + lr;
              const float v = Jr[col < NVP ? col : 0];
              av[t] = (rok && col < NVP) ? v : 0.f;
            }
            dsc = rok ? dl : 0.f;
          };
          float dsc, av[NT];
          fetch(0, dsc, av);
          for (int kb = 0; kb < K4; kb++) {
            float dsn, an[NT], bv[NT];
            fetch(kb + 1, dsn, an);
#pragma unroll
            for (int t = 0; t < NT; t++) bv[t] = av[t] * dsc;
#pragma unroll
            for (int ti = 0; ti < NT; ti++)
#pragma unroll
              for (int tj = ti; tj < NT; tj++) acc[ti][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ti], bv[tj], acc[ti][tj], 0, 0, 0);
            dsc = dsn;
#pragma unroll
            for (int t = 0; t < NT; t++) av[t] = an[t];
          }
#pragma unroll
          for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int tj = ti; tj < NT; tj++)
#pragma unroll
              for (int v = 0; v < 4; v++) {
                const int i = 16 * ti + 4 * lk + v, j = 16 * tj + lr;
                if (i < NVP && j < NVP) {
                  T[i * TD + j] = acc[ti][tj][v];
                  if (ti != tj) T[j * TD + i] = acc[ti][tj][v];
                }
              }
        } else {
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
          f4v acc[NT];
#pragma unroll
          for (int q = 0; q < NT; q++) acc[q] = f4v{0.f, 0.f, 0.f, 0.f};
          // K sweep, software-pipelined by hand: the operands of step kb + 1 are requested before the MFMAs of step kb issue, so
          // the LDS round trip (~100 cycles for a lone wave) hides behind the three 32-cycle matrix instructions instead of
          // preceding them.  Clamped addresses + unconditional loads + a select: `ok ? table[i] : 0` compiles to a branch around
          // the ds_read with a full s_waitcnt behind it.
          auto fetch = [&](const int kb, float& dsc, float (&av)[NT]) __attribute__((always_inline)) {
            const int row = 4 * kb + lk;
            const bool rok = row < erows;
            const int rr = rok ? row : 0;
            const float* Jr = Jb + rr * RS;
            const float dl = Dv[rr];
#pragma unroll
            for (int t = ti; t < NT; t++) {
              const int col = 16 * t + lr;
              const float v = Jr[col < NVP ? col : 0];
              av[t] = (rok && col < NVP) ? v : 0.f;
            }
            dsc = rok ? dl : 0.f;
          };
          float dsc, av[NT];
          fetch(0, dsc, av);
          for (int kb = 0; kb < K4; kb++) {
            float dsn, an[NT];
            fetch(kb + 1, dsn, an);
#pragma unroll
            for (int tj = ti; tj < NT; tj++) acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ti], av[tj] * dsc, acc[tj], 0, 0, 0);
            dsc = dsn;
#pragma unroll
            for (int t = ti; t < NT; t++) av[t] = an[t];
          }
#pragma unroll
          for (int tj = ti; tj < NT; tj++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
              const int i = 16 * ti + 4 * lk + v, j = 16 * tj + lr;
              if (i < NVP && j < NVP) {
                T[i * TD + j] = acc[tj][v];
                if (ti != tj) T[j * TD + i] = acc[tj][v];
              }
            }
        }
        }
        GSYNC();
        const int row = g < NVP ? g : 0;
#pragma unroll
        for (int k4 = 0; k4 < NVP / 4; k4++) {
          const float4 r = *reinterpret_cast<const float4*>(T + row * TD + 4 * k4);
          A[4 * k4] = Mrow[4 * k4] + r.x; A[4 * k4 + 1] = Mrow[4 * k4 + 1] + r.y;
          A[4 * k4 + 2] = Mrow[4 * k4 + 2] + r.z; A[4 * k4 + 3] = Mrow[4 * k4 + 3] + r.w;
        }
        if (g >= NVP) {
#pragma unroll
          for (int k = 0; k < NVP; k++) A[k] = 0.f;
        }
        GSYNC();                                  // the factor below rewrites the tile
      } else
#endif
      {
        // narrower groups (several envs per wave): lane i accumulates row i, the J row is an LDS broadcast
#pragma unroll
        for (int k = 0; k < NVP; k++) A[k] = Mrow[k];
        const float dr = on ? r_D : 0.f;
        const int col = g < NVP ? g : 0;
        for (int r = 0; r < nrows_wave; r++) {
          const float sD = bc<G>(dr, r);
          if (__ballot(sD != 0.f) == 0ull) continue;
          const float* Jr = W + KL().efcJ + r * RS;
          const float c = g < NVP ? sD * Jr[col] : 0.f;
          const float4* Jr4 = reinterpret_cast<const float4*>(Jr);
#pragma unroll
          for (int k = 0; k < NVP / 4; k++) {
            const float4 j4 = Jr4[k];
            A[4 * k] += c * j4.x; A[4 * k + 1] += c * j4.y; A[4 * k + 2] += c * j4.z; A[4 * k + 3] += c * j4.w;
          }
        }
      }
      PFN(PF_N_HBUILD, tn_);
      factor_core<false>(A);
      PFN(PF_N_FACTOR, tn_);
      float search = -solve(grad);
      if (g >= nv || done) search = 0.f;
      PFN(PF_N_SOLVE, tn_);
      float sn = sqrtf(gsum<G>(search * search));
      if (sn < MINVALF) done = true;
      float Mv = mul_m(search);
      float jv = jac_mul(search);
      PFN(PF_N_PROD, tn_);
      float dm = Ma - d_smooth;
      float q1 = gsum<G>(search * dm), q2 = gsum<G>(0.5f * search * Mv);
      const float gtol = KD().tolerance * KD().ls_tolerance * sn / scale;
      float alpha = 1.f, lo = 0.f, hi = -1.f;
      bool lsdone = done;
      for (int it = 0; it < KD().ls_iterations; it++) {
        float x = r_jar + alpha * jv;
        float d1 = 0.f, d2 = 0.f;
        {
          bool q;
          const float f = row_force(x, q);
          d1 = -f * jv;
          if (q) d2 = r_D * jv * jv;
        }
        d1 = gsum<G>(d1) + q1 + 2.f * alpha * q2;
        d2 = gsum<G>(d2) + 2.f * q2;
        if (!lsdone) {
          if (fabsf(d1) < fmaxf(gtol, 1e-6f * fabsf(q1))) lsdone = true;
          else {
            if (d1 < 0.f) lo = alpha; else hi = alpha;
            float next = alpha - d1 / fmaxf(d2, MINVALF);
            if (hi >= 0.f && (next <= lo || next >= hi)) next = 0.5f * (lo + hi);
            else if (hi < 0.f && next <= lo) next = 2.f * lo + 1e-10f;
            if (MM_LS_RELSTOP ? fabsf(next - alpha) <= MM_LS_RELSTOP_TOL * fabsf(alpha) : next == alpha) lsdone = true;
            alpha = next;
          }
        }
        if (__ballot(!lsdone) == 0ull) break;
      }
      if (!(alpha > 0.f)) done = true;
      if (!done) {
        d_qacc += alpha * search; Ma += alpha * Mv; r_jar += alpha * jv;
        alpha_prev = alpha;
        niter = iter + 1;
      }
      {
        float stepmax = gmax<G>(fabsf(alpha * search)), qmax = gmax<G>(fabsf(d_qacc));
        if (!done && stepmax <= 2e-7f * fmaxf(qmax, 1.f)) done = true;
      }
      if (iter == KD().iterations - 1 && !done) status |= 4;
      PFN(PF_N_LS, tn_);
    }
    bool on2;
    d_qfrccon = jacT_mul(row_force(r_jar, on2));
    PFN(PF_N_GRAD, tn_);
#undef PFN
  }

  // ------------------------------------------------------------------ pipeline
#define PFT(stage, call)                                   \
  do {                                                     \
    unsigned long long t0_ = (MM_STAGE_PROF && a.prof) ? clock64() : 0;       \
    MM_FENCE();                                            \
    call;                                                  \
    MM_FENCE();                                            \
    if (MM_STAGE_PROF && a.prof) pf[MM_STAGE_PROF ? stage : 0] += clock64() - t0_;              \
  } while (0)
  __device__ __forceinline__ void forward() {
    PFT(PF_KIN, kinematics());
    PFT(PF_COM, com_pos());
    const bool tw = TW && a.two_wave;
    if (tw) {
      tw_signal(0, ++tw_n);                       // poses are final: the helper wave starts on the tendons
      if (KD().ntlim) tw_wait(1, tw_n);           // tendon-limit rows need its lengths and Jacobian
    } else PFT(PF_TENDON, tendon());
    PFT(PF_CONSTR, make_constraint());
    PFT(PF_VEL, velocity_bias());
    PFT(PF_CRB, crb());
    if (tw && !SP && !IMPL && KD().any_damping && KD().eulerdamp) {   // M for the helper wave's Euler factor
      if (g < NVP) {
        float* Mg = W + KL().mtile + g * TD;
#pragma unroll
        for (int k4 = 0; k4 < NVP / 4; k4++)
          *reinterpret_cast<float4*>(Mg + 4 * k4) = make_float4(Mrow[4 * k4], Mrow[4 * k4 + 1], Mrow[4 * k4 + 2], Mrow[4 * k4 + 3]);
      }
      tw_signal(2, tw_n);
    }
    constexpr bool SPG = GEN && MM_SPARSE_LDL && MM_SPARSE_GEN && NVP >= 8 && INTEG != 2;
    const bool spg = SPG && (MM_SPARSE_GEN == 2 || KD().seg_nlevel > 0);   // general-row kernel on a model whose dof tree the sparse solve handles
    if constexpr (!SP) { if (!spg) PFT(PF_FACTOR, factor(0.f)); }
    PFT(PF_ACT, passive_actuation());
    if constexpr (SP) PFT(PF_SOLVE0, d_qaccsm = sp_factor_solve(0.f, d_smooth));
    else {
      if (spg) { if constexpr (SPG) PFT(PF_SOLVE0, d_qaccsm = spg_factor_solve(0.f, d_smooth)); }
      else PFT(PF_SOLVE0, d_qaccsm = solve(d_smooth));
    }
    PFT(PF_NEWTON, solve_constraints());
  }

  __device__ __forceinline__ bool bad_state(bool check_acc) {
    const auto& L = KL();
    int bad = 0;
    for (int i = g; i < KD().nq; i += G) bad |= !(fabsf(W[L.qpos + i]) < 1e10f);
    if (g < KD().nv) {
      bad |= !(fabsf(d_qvel) < 1e10f);
      if (check_acc) bad |= !(fabsf(d_qacc) < 1e10f);
    }
    return gor<G>(bad) != 0;
  }
  __device__ __forceinline__ void reset_data() {
    const auto& L = KL();
    for (int i = g; i < KD().nq; i += G) W[L.qpos + i] = MF_(QPOS0)[i];
    d_qvel = 0.f; d_warm = 0.f;
    if (g < KD().nv) W[L.qvel + g] = 0.f;
    for (int i = g; i < KD().na; i += G) W[L.act + i] = 0.f;
    GSYNC();
  }

  // A9 semi-implicit Euler with implicit joint damping
  __device__ __forceinline__ void euler(float& time) {
    const auto& L = KL();
    const float h = KD().timestep;
    d_warm = d_qacc;
    float qa_ = d_qacc;
    if (KD().any_damping && KD().eulerdamp) {
      if (TW && !SP && a.two_wave) {
        // the helper wave factorised M + h B while this wave was in Newton: fetch row g of L, solve
        tw_wait(3, tw_n);
        const float* Lg = W + KL().mtile + (g < NVP ? g : 0) * TD;
#pragma unroll
        for (int k4 = 0; k4 < NVP / 4; k4++) {
          const float4 r = *reinterpret_cast<const float4*>(Lg + 4 * k4);
          Lrow[4 * k4] = g < NVP ? r.x : 0.f; Lrow[4 * k4 + 1] = g < NVP ? r.y : 0.f;
          Lrow[4 * k4 + 2] = g < NVP ? r.z : 0.f; Lrow[4 * k4 + 3] = g < NVP ? r.w : 0.f;
        }
        d_dinv = g < NVP ? W[KL().mtile + NVP * TD + g] : 1.f;
        scale_rows();
        o_tile = KL().mtile;
        qa_ = solve(g < KD().nv ? d_smooth + d_qfrccon : 0.f);
        o_tile = KL().u1;
      } else qa_ = factor_solve(g < KD().nv ? h * MF_(DOF_DAMPING)[g] : 0.f, g < KD().nv ? d_smooth + d_qfrccon : 0.f);
    }
    for (int u = g; u < KD().nu; u += G) {
      int aa = MI_(ACT_ACTADR)[u];
      const int dt_ = GEN ? MI_(ACT_DYNTYPE)[u] : 0;     // (general-row kernels: both words at once)
      if (aa < 0) continue;
      float x = W[L.act + aa] + h * W[L.actdot + aa];
      if ((GEN ? dt_ : MI_(ACT_DYNTYPE)[u]) == MM_DYN_MUSCLE) x = clampf(x, 0.f, 1.f);
      W[L.act + aa] = x;
    }
    if (g < KD().nv) {
      d_qvel += h * qa_;
      W[L.qvel + g] = d_qvel;
    }
    GSYNC();
    integrate_pos(L.qvel, h);
    time += h;
    GSYNC();
  }

  // mjINT_IMPLICITFAST (oracle: mmo_implicitfast): (M + h W) qacc* = qfrc_smooth + qfrc_constraint with
  // W = diag(dofw) + sum_t tenw_t J_t'J_t restricted to dof pairs on one kinematic chain (the pattern of M), then mj_advance.
  // Lane i builds row i of W in its own row of the dense LDS tile (the factor of M in there is dead once Newton is done).
  // W's tendon part sum_t w_t J_t'J_t (on-chain pairs) into the dense tile at word offset o_t: LDS in (tenw from the actuation
  // stage, the tendon Jacobian), LDS out -- a two-wave launch runs it in the helper wave while the main wave is in Newton
  __device__ __forceinline__ void implicit_w(int o_t) {
    const auto& L = KL();
    float* T = W + o_t;
    const int row = g < NVP ? g : 0;
    if (g < NVP)
#pragma unroll
      for (int k4 = 0; k4 < NVP / 4; k4++) *reinterpret_cast<float4*>(T + row * TD + 4 * k4) = make_float4(0.f, 0.f, 0.f, 0.f);
    GSYNC();
    {
      // one lane per tendon: its (<= 8 x 8) on-chain entry pairs go into the tile with LDS float atomics (a lane per dof walking
      // every tendon that crosses it -- ~20 for a hip dof -- serialises ~160 dependent read-modify-writes)
      int s_ja = SECOFF_(TENJ_ADR), s_jd = SECOFF_(TENJ_DOF), o_tj = L.tenj, o_tw = L.tenw, x_rel = KX().dof_rel;
      PIN_S(s_ja); PIN_S(s_jd); PIN_S(o_tj); PIN_S(o_tw); PIN_S(x_rel);
      for (int t = g; t < KD().ntendon; t += G) {
        const float wt = W[o_tw + t];
        if (wt == 0.f) continue;
        const int e0 = AI_(s_ja)[t], e1 = AI_(s_ja)[t + 1];
        for (int c0 = e0; c0 < e1; c0 += 8) {          // entries in chunks of eight held in registers: the pair loop below is
          int dd[8]; float jj[8]; unsigned rl[8], rh[8];   // then pure arithmetic + atomics, no load in its dependency chain
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const bool ok = c0 + k < e1;
            dd[k] = ok ? AI_(s_jd)[c0 + k] : -1;
            jj[k] = ok ? W[o_tj + c0 + k] : 0.f;
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int d1 = dd[k] < 0 ? 0 : dd[k];
            rl[k] = (unsigned)AI_(x_rel)[2 * d1]; rh[k] = (unsigned)AI_(x_rel)[2 * d1 + 1];
          }
          for (int c1 = e0; c1 < e1; c1 += 8) {
            int d2[8]; float j2[8];
            if (c1 == c0) {
#pragma unroll
              for (int k = 0; k < 8; k++) { d2[k] = dd[k]; j2[k] = jj[k]; }
            } else {
#pragma unroll
              for (int k = 0; k < 8; k++) {
                const bool ok = c1 + k < e1;
                d2[k] = ok ? AI_(s_jd)[c1 + k] : -1; j2[k] = ok ? W[o_tj + c1 + k] : 0.f;
              }
            }
#pragma unroll
            for (int a_ = 0; a_ < 8; a_++) {
              if (dd[a_] < 0) continue;
              const float w1 = wt * jj[a_];
#pragma unroll
              for (int b_ = 0; b_ < 8; b_++) {
                if (d2[b_] < 0) continue;
                const bool rel = d2[b_] < 32 ? (rl[a_] >> d2[b_]) & 1u : (rh[a_] >> (d2[b_] - 32)) & 1u;
                if (rel) atomicAdd(&T[dd[a_] * TD + d2[b_]], w1 * j2[b_]);
              }
            }
          }
        }
      }
    }
    GSYNC();
  }
  __device__ __forceinline__ void implicit_step(float& time) {
    const auto& L = KL();
    const float h = KD().timestep;
    const int nv = KD().nv;
    d_warm = d_qacc;
    const bool twi = TW && a.two_wave;
    if (twi) tw_wait(3, tw_n);          // the helper wave assembled W's tendon part in the second tile
    else implicit_w(L.u1);
    float* T = W + (twi ? L.mtile : L.u1);
    const int row = g < NVP ? g : 0;
    float A[NVP];
#pragma unroll
    for (int k4 = 0; k4 < NVP / 4; k4++) {
      const float4 r = *reinterpret_cast<const float4*>(T + row * TD + 4 * k4);
      A[4 * k4] = Mrow[4 * k4] + h * r.x; A[4 * k4 + 1] = Mrow[4 * k4 + 1] + h * r.y;
      A[4 * k4 + 2] = Mrow[4 * k4 + 2] + h * r.z; A[4 * k4 + 3] = Mrow[4 * k4 + 3] + h * r.w;
    }
    if (g >= NVP) {
#pragma unroll
      for (int k = 0; k < NVP; k++) A[k] = 0.f;
    }
    const float dw = g < nv ? h * W[L.dofw + g] : 0.f;
#pragma unroll
    for (int k = 0; k < NVP; k++) A[k] += (k == g) ? dw : 0.f;
    GSYNC();
    factor_core<false>(A);
    const float qa_ = solve(g < nv ? d_smooth + d_qfrccon : 0.f);
    for (int u = g; u < KD().nu; u += G) {
      int aa = MI_(ACT_ACTADR)[u];
      if (aa < 0) continue;
      float x = W[L.act + aa] + h * W[L.actdot + aa];
      if (MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) x = clampf(x, 0.f, 1.f);
      W[L.act + aa] = x;
    }
    if (g < nv) {
      d_qvel += h * qa_;
      W[L.qvel + g] = d_qvel;
    }
    GSYNC();
    integrate_pos(L.qvel, h);
    time += h;
    GSYNC();
  }

  // qpos <- qpos (+) hh * vel on the configuration manifold (mj_integratePos); vel = LDS vector at word offset `voff`
  __device__ __forceinline__ void integrate_pos(int voff, float hh) {
    // offsets read once and pinned in SGPRs for this stage (see PIN_S)
    int o_qpos = KL().qpos; PIN_S(o_qpos); int s_JNT_TYPE = SECOFF_(JNT_TYPE); PIN_S(s_JNT_TYPE); int s_JNT_QPOSADR = SECOFF_(JNT_QPOSADR); PIN_S(s_JNT_QPOSADR); int s_JNT_DOFADR = SECOFF_(JNT_DOFADR); PIN_S(s_JNT_DOFADR); int d_njnt_ = KD().njnt; PIN_S(d_njnt_);
    const auto& L = KL();
    for (int j = g; j < d_njnt_; j += G) {
      int type = AI_(s_JNT_TYPE)[j], qa = AI_(s_JNT_QPOSADR)[j], da = AI_(s_JNT_DOFADR)[j];
      if (type == MM_JNT_HINGE || type == MM_JNT_SLIDE) { W[o_qpos + qa] += hh * W[voff + da]; continue; }
      if (type == MM_JNT_FREE) {
        for (int k = 0; k < 3; k++) W[o_qpos + qa + k] += hh * W[voff + da + k];
        qa += 3; da += 3;
      }
      V3 w = ld3(W + voff + da);
      float nw = sqrtf(dot(w, w)), ang = hh * nw;
      if (ang > MINVALF) {
        float sn, cs;
        sincos_small(0.5f * ang, &sn, &cs);
        float is = sn / nw;
        Q4 dq = {cs, w.x * is, w.y * is, w.z * is};
        Q4 qn = qnorm(qmul(ldq(W + o_qpos + qa), dq));
        W[o_qpos + qa] = qn.w; W[o_qpos + qa + 1] = qn.x; W[o_qpos + qa + 2] = qn.y; W[o_qpos + qa + 3] = qn.z;
      }
    }
  }

  // One stage of classical RK4 (mj_RungeKutta, N = 4; oracle: mmo_rk4).  Called after the forward pass of stage `i`
  // (i = 0 is mj_step's own forward).  Stages 0..2 move the state to X0 + h a_i F_i; stage 3 applies the weighted update.
  __device__ __forceinline__ void rk4_stage(int i, float& time, float t0) {
    const auto& L = KL();
    const float h = KD().timestep;
    const float A_ = i == 2 ? 1.f : 0.5f;
    const float B_ = (i == 0 || i == 3) ? (1.f / 6.f) : (1.f / 3.f);
    d_warm = d_qacc;
    if (i == 0) {
      rk_v0 = d_qvel; rk_vsum = 0.f; rk_asum = 0.f;
      for (int k = g; k < KD().nq; k += G) W[L.rk_qpos0 + k] = W[L.qpos + k];
      for (int k = g; k < KD().na; k += G) { W[L.rk_act0 + k] = W[L.act + k]; W[L.rk_adot + k] = 0.f; }
    }
    rk_vsum += B_ * d_qvel; rk_asum += B_ * d_qacc;
    for (int k = g; k < KD().na; k += G) W[L.rk_adot + k] += B_ * W[L.actdot + k];
    GSYNC();
    // velocity used for the position update of this stage goes through the (free) L.vec scratch vector
    const float hh = i < 3 ? h * A_ : h;
    if (g < KD().nv) W[L.vec + g] = i < 3 ? d_qvel : rk_vsum;
    for (int k = g; k < KD().nq; k += G) W[L.qpos + k] = W[L.rk_qpos0 + k];
    GSYNC();
    integrate_pos(L.vec, hh);
    if (g < KD().nv) {
      d_qvel = i < 3 ? rk_v0 + hh * d_qacc : rk_v0 + h * rk_asum;
      W[L.qvel + g] = d_qvel;
    }
    for (int u = g; u < KD().nu; u += G) {
      int aa = MI_(ACT_ACTADR)[u];
      if (aa < 0) continue;
      float x = i < 3 ? W[L.rk_act0 + aa] + hh * W[L.actdot + aa] : W[L.rk_act0 + aa] + h * W[L.rk_adot + aa];
      if (i == 3 && MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) x = clampf(x, 0.f, 1.f);
      W[L.act + aa] = x;
    }
    time = i < 3 ? t0 + hh : t0 + h;
    GSYNC();
  }

  // `nsub` mj_step substeps (forward + Euler, MuJoCo bad-state auto-reset semantics) followed by an
  // optional mj_forward on the final state.  One call site of forward() keeps the code size bounded.
  __device__ __forceinline__ void run(int nsub, bool final_forward, float& time) {
    int total = nsub + (final_forward ? 1 : 0);
    int s = 0;
    bool redo = false;
    if constexpr (!RK4) {
      while (s < total) {
        const bool stepping = s < nsub;
        if (stepping && !redo && bad_state(false)) { reset_data(); time = 0.f; status |= 1; }
        forward();
        if (stepping) {
          if (!redo && bad_state(true)) {
            // two-wave launches: the helper may still be factorising the aborted pass's M + h B (implicitfast: assembling W) in
            // the second tile, which the redone forward pass rewrites -- let it finish first
            if constexpr (TW) { if (a.two_wave && (IMPL || (!SP && KD().any_damping && KD().eulerdamp))) tw_wait(3, tw_n); }
            reset_data(); time = 0.f; status |= 1; redo = true; continue;
          }
          if constexpr (IMPL) { PFT(PF_EULER, implicit_step(time)); }
          else { PFT(PF_EULER, euler(time)); }
          redo = false;
        }
        s++;
      }
    } else {
      // RK4 is a compile-time variant: its stage machine costs the Euler kernels registers if it shares their code
      int rk = 0;
      float t0 = time;
      while (s < total) {
        const bool stepping = s < nsub;
        if (stepping && rk == 0 && !redo && bad_state(false)) { reset_data(); time = 0.f; status |= 1; }
        forward();
        if (stepping) {
          if (rk == 0 && !redo && bad_state(true)) { reset_data(); time = 0.f; status |= 1; redo = true; continue; }
          if (rk == 0) t0 = time;
          PFT(PF_EULER, rk4_stage(rk, time, t0));
          rk = (rk + 1) & 3;
          if (rk == 0) { redo = false; s++; }
          continue;
        }
        s++;
      }
    }
  }
};

// =========================================================================== kernels
// OBS: the reset-observation pass (mm_task.obs_only: forward pass + observation for the envs of a mask, no stepping) as its own
// kernel symbol for the workloads whose reset is not folded into the env-step launch -- rocprofv3's per-kernel statistics then
// separate the env-step from the (mostly early-exiting) pass that follows it, and the pass sheds the integrator code.
template <int G, int NVP, bool LM, bool GEN, int INTEG, bool OBS = false>
__global__ void __launch_bounds__(512) k_engine(KArgs a) {
  extern __shared__ float lds[];
  constexpr int EPW = 64 / G;  // envs per wave
  const int lane = threadIdx.x & 63;
  // two-wave launches (Engine::TW): waves [0, wpb) are the main waves of the block's envs, waves [wpb, 2 wpb) their helpers
  const bool two_wave = Engine<G, NVP, GEN, INTEG>::TW && a.two_wave;
  const int wpb = two_wave ? (blockDim.x >> 7) : (blockDim.x >> 6);
  const bool helper = two_wave && (int)(threadIdx.x >> 6) >= wpb;
  const int wave = (int)(threadIdx.x >> 6) - (helper ? wpb : 0);
  const int g = lane % G;
  unsigned long long t_start = (MM_STAGE_PROF && a.prof) ? clock64() : 0;
  // reset-observation pass (mm_task.obs_only with an env mask): a block none of whose envs is flagged leaves before the model
  // is staged -- every wave scans the block's whole env range, so the decision is block-uniform and nobody is left waiting at
  // the barrier (the pass is launched after every step of the non-Pose tasks and usually has nothing to do)
  if (a.mode == 2 && (OBS || KA().t.obs_only) && KA().t.env_mask) {
    const int epb = wpb * EPW, e0 = blockIdx.x * epb;
    bool any = false;
    for (int i = lane; i < epb; i += 64) any |= (e0 + i < a.s.nenv) && KA().t.env_mask[e0 + i] != 0;
    if (__ballot(any) == 0ull) return;
  }
  // ---- stage the model tables into LDS once per block (all waves participate)
  const uint32_t* mb = a.blob;
  float* wsbase = lds;
  if (LM) {
    uint32_t* lm = reinterpret_cast<uint32_t*>(lds);
    // 128-bit copies, four in flight per thread (word by word this was one serialised HBM / L2 round trip per 2 KB of model: 16-22
    // of them, ~1 % of the launch)
    const uint4* src4 = reinterpret_cast<const uint4*>(a.blob);
    uint4* dst4 = reinterpret_cast<uint4*>(lm);
    const int n4 = a.blob_words >> 2;
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += blockDim.x) dst4[i] = src4[i];
    for (int i = (n4 << 2) + threadIdx.x; i < a.blob_words; i += blockDim.x) lm[i] = a.blob[i];
    __syncthreads();
    mb = lm;
    wsbase = lds + ((a.blob_words + 3) & ~3);
  }
  if (two_wave) {   // the meeting counters of the block's envs start at zero
    if ((int)threadIdx.x < wpb * EPW) {
      float* Wf = wsbase + (size_t)threadIdx.x * KL().total + KL().flags;
      reinterpret_cast<int*>(Wf)[0] = 0; reinterpret_cast<int*>(Wf)[1] = 0; reinterpret_cast<int*>(Wf)[2] = 0; reinterpret_cast<int*>(Wf)[3] = 0;
    }
    __syncthreads();
  }
  KConst kc;
#if MM_CONST_IN_REGS
  {
    ConstWords hdr = reinterpret_cast<ConstWords>(reinterpret_cast<uintptr_t>(a.blob));
    ConstWords cbw = reinterpret_cast<ConstWords>(reinterpret_cast<uintptr_t>(a.blob + a.cofs));
    static_assert(sizeof(ConstBlock) == sizeof(Dims) + sizeof(Layout) + sizeof(Aux) && sizeof(ConstBlock) % 4 == 0, "ConstBlock is three packed word structs");
    uint32_t* dst = reinterpret_cast<uint32_t*>(&kc);   // KConst starts with {Dims, Layout, Aux} = the ConstBlock
#pragma unroll
    for (int w_ = 0; w_ < (int)(sizeof(ConstBlock) / 4); w_++) dst[w_] = cbw[w_];
#if MM_SEC_IN_REGS
#pragma unroll
    for (int s_ = 0; s_ < MM_NSEC; s_++) kc.sec[s_] = (int)hdr[MM_HEADER_WORDS + 2 * s_];
#endif
  }
#endif
  int e = (blockIdx.x * wpb + wave) * EPW + lane / G;
  const int nenv = a.s.nenv;
  if ((blockIdx.x * wpb + wave) * EPW >= nenv) return;  // whole wave idle
  bool dup = e >= nenv;
  if (dup) e = nenv - 1;  // surplus groups recompute the last env (they never store)
  if (a.mode == 2 && KA().t.env_mask && !KA().t.env_mask[e]) dup = true;   // masked-out envs are left untouched
  bool obs_only = OBS || (a.mode == 2 && KA().t.obs_only);
  if (obs_only && __ballot(!dup) == 0ull) return;   // reset-observation pass: waves without a reset env do nothing
  float* W = wsbase + (size_t)(wave * EPW + lane / G) * KL().total;
  const auto& L = KL();
  const auto& d = KD();
  Engine<G, NVP, GEN, INTEG> E(a, kc, mb, W, g);
  if (a.s.geom_size_env && a.s.geom_env_id >= 0) {
    E.env_has_gs = true;
    E.env_gsv[0] = a.s.geom_size_env[(size_t)e * 3]; E.env_gsv[1] = a.s.geom_size_env[(size_t)e * 3 + 1]; E.env_gsv[2] = a.s.geom_size_env[(size_t)e * 3 + 2];
  }
  if (a.s.geom_type_env && a.s.geom_env_id >= 0) E.env_gtype = a.s.geom_type_env[e];
  E.env = e;
  if constexpr (Engine<G, NVP, GEN, INTEG>::TW) {
    if (helper) { E.helper_loop(); return; }   // everything it needs and leaves lives in the env's LDS tables
  }

  // ---- load state (HBM -> LDS tables / owner registers)
  for (int i = g; i < d.nq; i += G) W[L.qpos + i] = a.s.qpos[(size_t)e * d.nq + i];
  if (g < d.nv) {
    E.d_qvel = a.s.qvel[(size_t)e * d.nv + g];
    E.d_warm = a.s.qacc_warmstart[(size_t)e * d.nv + g];
    W[L.qvel + g] = E.d_qvel;
  }
  for (int i = g; i < d.na; i += G) W[L.act + i] = a.s.act[(size_t)e * d.na + i];
  float time = a.s.time[e];
  E.status = a.s.status ? a.s.status[e] : 0;
  const __attribute__((address_space(4))) mm_task& t = KA().t;
  // ---- action -> ctrl (BaseV0.step: base_v0.py:82-108)
  const bool has_ro = a.mode == 2 && a.has_ro && !obs_only;
  for (int u = g; u < d.nu; u += G) {
    float c = a.ctrl ? a.ctrl[(size_t)e * d.nu + u] : 0.f;
    if (has_ro && !a.ctrl) {
      // action ~ U[0,1) drawn here (benchmarks/mjx_benchmark.py:29), element-for-element what mm_uniform writes for the flat
      // index of (global env, actuator)
      const uint64_t i = (uint64_t)(a.s.env_index_base + e) * (uint64_t)d.nu + (uint64_t)u, i4 = i >> 2;
      const uint64_t sd = KA().ro.action_seed, sm = KA().ro.action_stream;
      uint32_t cc[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)sm, (uint32_t)(sm >> 32)};
      philox4x32_10(cc, (uint32_t)sd, (uint32_t)(sd >> 32));
      const int w = (int)(i & 3);
      c = u01(w == 0 ? cc[0] : (w == 1 ? cc[1] : (w == 2 ? cc[2] : cc[3])));
      if (KA().ro.action_out && !dup) KA().ro.action_out[(size_t)e * d.nu + u] = c;
    }
    const bool mus = MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE;
    if (obs_only) c = 0.f;
    if (a.mode == 2 && !obs_only && t.normalize_act && mus) c = 1.f / (1.f + expf(-5.f * (c - 0.5f)));
    // no activation states at all (motorFinger): BaseV0.step hands the normalisation to the robot, which maps [-1, 1] onto
    // the ctrl range (base_v0.py:94-96, robot.py:786-796)
    if (a.mode == 2 && !obs_only && t.normalize_act && d.na == 0)
      c = 0.5f * (MF_(ACT_CTRLRANGE)[2 * u] + MF_(ACT_CTRLRANGE)[2 * u + 1]) +
          c * 0.5f * (MF_(ACT_CTRLRANGE)[2 * u + 1] - MF_(ACT_CTRLRANGE)[2 * u]);
    if (a.mode == 2 && !obs_only && t.fatigue && mus) {
      // 3CC-r muscle fatigue (fatigue.py:38-76), dt = timestep * frame_skip
      int aa = MI_(ACT_ACTADR)[u];
      size_t k = (size_t)e * d.na + aa;
      float MA = t.fat_MA[k], MR = t.fat_MR[k], MF = t.fat_MF[k], TL = c;
      float dt = d.timestep * (float)t.nsubsteps;
      float tauact = MF_(ACT_DYNPRM)[3 * u], taudeact = MF_(ACT_DYNPRM)[3 * u + 1];
      float LD = 1.f / tauact * (0.5f + 1.5f * MA), LR = (0.5f + 1.5f * MA) / taudeact;
      float C, rR;
      if (MA < TL) { C = MR > (TL - MA) ? LD * (TL - MA) : LD * MR; rR = t.fat_R; }
      else { C = LR * (TL - MA); rR = t.fat_r * t.fat_R; }
      float lo = fmaxf(-MA / dt + t.fat_F * MA, (MR - 1.f) / dt + rR * MF);
      float hi = fminf((1.f - MA) / dt + t.fat_F * MA, MR / dt + rR * MF);
      C = fminf(fmaxf(C, lo), hi);
      float dMA = (C - t.fat_F * MA) * dt, dMR = (-C + rR * MF) * dt, dMF = (t.fat_F * MA - rR * MF) * dt;
      MA += dMA; MR += dMR; MF += dMF;
      if (!dup) { t.fat_MA[k] = MA; t.fat_MR[k] = MR; t.fat_MF[k] = MF; }
      c = MA;
    }
    W[L.ctrl + u] = c;
  }
  GSYNC();
  if (a.mode == 2 && !obs_only && t.reaf_src >= 0 && t.reaf_dst >= 0 && g == 0) {  // base_v0.py:104-108
    W[L.ctrl + t.reaf_dst] = W[L.ctrl + t.reaf_src];
    W[L.ctrl + t.reaf_src] = 0.f;
  }
  GSYNC();
  if (a.mode == 2 && t.ctrl_out && !dup)
    for (int u = g; u < d.nu; u += G) t.ctrl_out[(size_t)e * d.nu + u] = W[L.ctrl + u];

  int nsub = (OBS || a.mode == 1 || obs_only) ? 0 : t.nsubsteps;
  bool fwd = OBS || a.mode == 1 || obs_only || (a.mode == 2 && t.do_forward);
  // FOLD: the masked auto-reset of the WALK / REORIENT tasks inside this launch (mm_rollout.autoreset).  Their first observation
  // needs a forward pass on the reset state: an env that ends its episode is re-armed in registers / LDS after the task stage and
  // the same wave runs the reset-observation pass (forward + observation, no stepping, no bookkeeping) before the state is stored.
  // One env per wavefront (G = 64), so the second pass is a wave-uniform branch.  It is a SECOND inlined copy of the forward
  // pipeline and of the task stage (forward only, obs_only constant), not a loop around one copy: the loop kept everything the
  // pipeline reads live across the task stage and the reset block (100+ VGPR spills in the 32- / 36-wide kernels, kernel time
  // +1..3 %); the straight-line form has the spill count and the kernel time of the unfolded kernel (0 / 0 / 5 spills), and the
  // cold copy is only fetched by a wave whose env resets.  Compiled into the kernels that also exist as reset-observation kernels
  // (MM_KERNELS_OBS: the 32- and 36-wide ones of the reorient and leg models) -- the only models the two tasks run on.
  constexpr bool FOLD = MM_FOLD_RESET && GEN && G == 64 && !OBS && (NVP == 32 || NVP == 36);
  bool refold = false;
  // results of the task stage the rollout bookkeeping needs: dense reward / solved / done (valid in lane 0 of the group),
  // and whether this env is re-armed inside this launch (group-uniform; POSE task with mm_rollout.autoreset)
  float rw_dense = 0.f, rw_solved = 0.f;
  bool rw_done = false, will_reset = false;
  E.run(nsub, fwd, time);

  // everything between the pipeline and the state store: derived outputs, task stage, bookkeeping.  `pass` 1 = the reset-observation
  // pass of a folded reset (observation only: obs_only is true there)
  auto stage = [&](const bool obs_only, const int pass) __attribute__((always_inline)) {
  // ---- derived outputs of the final forward
  if (fwd && a.has_derived) {
    const mm_derived& o = a.o;
    const bool isb = g < d.nbody;
    const V3 org = E.origin();   // outputs are world coordinates
    if (o.xpos && isb) st3(o.xpos + ((size_t)e * d.nbody + g) * 3, E.b_xpos + org);
    if (o.xquat && isb) { float* q = o.xquat + ((size_t)e * d.nbody + g) * 4; q[0] = E.b_xquat.w; q[1] = E.b_xquat.x; q[2] = E.b_xquat.y; q[3] = E.b_xquat.z; }
    if (o.xipos && isb) st3(o.xipos + ((size_t)e * d.nbody + g) * 3, E.b_xipos + org);
    if (o.cvel && isb) for (int k = 0; k < 6; k++) o.cvel[((size_t)e * d.nbody + g) * 6 + k] = E.b_cvel[k];
    if (o.subtree_com && isb) st3(o.subtree_com + ((size_t)e * d.nbody + g) * 3, ld3(W + L.com + 3 * AUXI(body_rootslot)[g]) + org);
    if (o.site_xpos)
      for (int s = g; s < d.nsite; s += G) st3(o.site_xpos + ((size_t)e * d.nsite + s) * 3, E.site_pos(s) + org);
    if (o.geom_xpos)
      for (int s = g; s < d.ngeom; s += G) st3(o.geom_xpos + ((size_t)e * d.ngeom + s) * 3, E.geom_pos(s) + org);
    if (o.actuator_length) for (int i = g; i < d.nu; i += G) o.actuator_length[(size_t)e * d.nu + i] = W[L.actlen + i];
    if (o.actuator_velocity) for (int i = g; i < d.nu; i += G) o.actuator_velocity[(size_t)e * d.nu + i] = W[L.actvel + i];
    if (o.actuator_force) for (int i = g; i < d.nu; i += G) o.actuator_force[(size_t)e * d.nu + i] = W[L.actfrc + i];
    if (o.qacc && g < d.nv) o.qacc[(size_t)e * d.nv + g] = E.d_qacc;
    if (o.ten_length) for (int i = g; i < d.ntendon; i += G) o.ten_length[(size_t)e * d.ntendon + i] = W[L.tenlen + i];
    if (g == 0 && o.nefc) o.nefc[e] = E.nefc;
    if (g == 0 && o.solver_niter) o.solver_niter[e] = E.niter;
  }
  if (a.dbg) {  // tests only: owner registers and tables in a flat record
    float* D = a.dbg + (size_t)e * a.D.total;
    if constexpr (Engine<G, NVP, GEN, INTEG>::SP) {
      E.sp_dense_tile();
      if (g < d.nv) for (int k = 0; k < d.nv; k++) D[a.D.M + g * d.nv + k] = W[L.u1 + g * Engine<G, NVP, GEN, INTEG>::TD + k];
    }
    if (g < d.nbody) {
      st3(D + a.D.xpos + 3 * g, E.b_xpos + E.origin()); st3(D + a.D.xipos + 3 * g, E.b_xipos + E.origin());
      D[a.D.xquat + 4 * g] = E.b_xquat.w; D[a.D.xquat + 4 * g + 1] = E.b_xquat.x;
      D[a.D.xquat + 4 * g + 2] = E.b_xquat.y; D[a.D.xquat + 4 * g + 3] = E.b_xquat.z;
      for (int k = 0; k < 6; k++) D[a.D.cvel + 6 * g + k] = E.b_cvel[k];
    }
    if (g < d.nv) {
      for (int k = 0; k < 6; k++) D[a.D.cdof + 6 * g + k] = E.d_cdof[k];
      if constexpr (!Engine<G, NVP, GEN, INTEG>::SP) {
#pragma unroll
        for (int k = 0; k < NVP; k++) if (k < d.nv) D[a.D.M + g * d.nv + k] = E.Mrow[k];
      }
      D[a.D.bias + g] = E.d_bias; D[a.D.smooth + g] = E.d_smooth; D[a.D.qaccsm + g] = E.d_qaccsm;
      D[a.D.qacc + g] = E.d_qacc; D[a.D.qfrccon + g] = E.d_qfrccon;
    }
    for (int i = g; i < d.ntendon; i += G) { D[a.D.tenlen + i] = W[L.tenlen + i]; D[a.D.tenvel + i] = W[L.tenvel + i]; }
    for (int i = g; i < d.ntenJ; i += G) D[a.D.tenj + i] = W[L.tenj + i];
    for (int i = g; i < d.nu; i += G) D[a.D.actfrc + i] = W[L.actfrc + i];
    for (int i = g; i < d.na; i += G) D[a.D.actdot + i] = W[L.actdot + i];
    D[a.D.efc_active + g] = E.r_active ? 1.f : 0.f; D[a.D.efc_D + g] = E.r_D; D[a.D.efc_aref + g] = E.r_aref;
    if (g == 0) D[a.D.scal] = (float)E.niter;
    if (g < 15) { D[a.D.scal + 1 + g] = E.r_jar; D[a.D.scal + 16 + g] = E.r_floss; }   // rows of small test models
  }
  if (MM_STAGE_PROF && a.prof && blockIdx.x == 0 && threadIdx.x == 0) {
    E.pf[MM_STAGE_PROF ? PF_TOTAL : 0] = clock64() - t_start;
#pragma unroll
    for (int i = 0; i < (MM_STAGE_PROF ? NPROF : 1); i++) a.prof[i] = E.pf[i];
  }

  // ---- task stage: obs_dict / reward_dict (pose_v0.py:100-140), TimeLimit counter
  if (a.mode == 2) {
    int sc = 0, sc0 = 0;
    if (t.step_count) { sc0 = (FOLD && pass == 1) ? 0 : t.step_count[e]; sc = obs_only ? sc0 : sc0 + 1; }
    if (t.task == MM_TASK_POSE) {
      const float dt = t.obs_dt;
      const int o_err = t.obs_layout == 1 ? d.nq + d.nv + d.na : d.nq + d.nv;
      const int o_act = t.obs_layout == 1 ? d.nq + d.nv : 2 * d.nq + d.nv;
      float err2 = 0.f, act2 = 0.f;
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      for (int i = g; i < d.nq; i += G) {
        const float pe = t.target_jnt_value[(size_t)e * d.nq + i] - W[L.qpos + i];
        err2 += pe * pe;
      }
      for (int i = g; i < d.na; i += G) { const float x = W[L.act + i]; act2 += x * x; }
      err2 = gsum<G>(err2); act2 = gsum<G>(act2);
      // group-uniform (gsum is bitwise uniform): every lane knows whether the episode ends here
      const float pose_dist = sqrtf(err2);
      const bool done = pose_dist > t.far_th;
      rw_done = done;
      will_reset = has_ro && KA().ro.autoreset && (done || (t.max_episode_steps > 0 && sc >= t.max_episode_steps));
      if (ob && !will_reset) {      // an env that resets in this launch gets the first observation of its new episode instead
        for (int i = g; i < d.nq; i += G) {
          const float q = W[L.qpos + i];
          ob[i] = q; ob[o_err + i] = t.target_jnt_value[(size_t)e * d.nq + i] - q;
        }
        if (g < d.nv) ob[d.nq + g] = E.d_qvel * dt;
        for (int i = g; i < d.na; i += G) ob[o_act + i] = W[L.act + i];
      }
      if (g == 0) {
        float act_mag = sqrtf(act2);
        if (d.na != 0 && t.act_reg_mean) act_mag = act_mag / (float)d.na;
        float r_pose = -pose_dist;
        float r_bonus = (pose_dist < t.pose_thd ? 1.f : 0.f) + (pose_dist < 1.5f * t.pose_thd ? 1.f : 0.f);
        float r_pen = pose_dist > t.far_th ? -1.f : 0.f;
        float r_act = -act_mag;
        rw_dense = t.w_pose * r_pose + t.w_bonus * r_bonus + t.w_act_reg * r_act + t.w_penalty * r_pen;
        rw_solved = pose_dist < t.pose_thd ? 1.f : 0.f;
        if (t.rwd && !obs_only) {   // the reset observation leaves the terminal step's reward terms in place
          float* r = t.rwd + (size_t)e * MM_RWD_COUNT;
          r[MM_RWD_POSE] = r_pose; r[MM_RWD_BONUS] = r_bonus; r[MM_RWD_PENALTY] = r_pen; r[MM_RWD_ACT_REG] = r_act;
          r[MM_RWD_SPARSE] = -pose_dist; r[MM_RWD_SOLVED] = rw_solved;
          r[MM_RWD_DONE] = done ? 1.f : 0.f;
          r[MM_RWD_DENSE] = rw_dense;
        }
        if (t.done && !obs_only) t.done[e] = done ? 1 : 0;
      }
    }
    if (t.task == MM_TASK_REACH) {
      // obs [qpos, qvel*dt, tip_pos, reach_err, act]; reward dict of reach_v0.py:123-151
      const int n3 = 3 * t.ntip;
      // obs_layout 1 = MJX order [qpos, qvel, act, tip_pos, reach_err] (playground_reach_v0.py:150-165)
      const int o_tip = t.obs_layout == 1 ? d.nq + d.nv + d.na : d.nq + d.nv;
      const int o_ract = t.obs_layout == 1 ? d.nq + d.nv : d.nq + d.nv + 2 * n3;
      float err2 = 0.f, act2 = 0.f;
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      for (int i = g; i < d.nq; i += G) if (ob) ob[i] = W[L.qpos + i];
      if (ob && g < d.nv) ob[d.nq + g] = E.d_qvel * t.obs_dt;
      const float vs = g < d.nv ? E.d_qvel * t.obs_dt : 0.f;
      const float vel2 = t.reach_stand ? gsum<G>(vs * vs) : 0.f;
      for (int i = g; i < t.ntip; i += G) {
        const V3 tip_i = E.site_pos(t.tip_sites[i]);
        V3 tip = tip_i + E.origin();
        V3 tgt = ld3(t.target_pos + (size_t)e * n3 + 3 * i);
        V3 er = (tgt - E.origin()) - tip_i;
        err2 += dot(er, er);
        if (ob) { st3(ob + o_tip + 3 * i, tip); st3(ob + o_tip + n3 + 3 * i, er); }
      }
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[o_ract + i] = x;
      }
      err2 = gsum<G>(err2); act2 = gsum<G>(act2);
      if (g == 0) {
        float reach_dist = sqrtf(err2), act_mag = d.na != 0 ? sqrtf(act2) / (float)d.na : 0.f;
        float far_th = time > 2.f * t.obs_dt ? t.reach_far_th * (float)t.ntip : INFINITY;
        float near_th = (float)t.ntip * (t.reach_stand ? 0.050f : 0.0125f);
        float r_reach = -reach_dist;
        if (t.reach_stand) { r_reach = 10.f - reach_dist - 10.f * sqrtf(vel2); act_mag *= 100.f; }   // walk_v0.py:100-111
        float r_bonus = (reach_dist < 2.f * near_th ? 1.f : 0.f) + (reach_dist < near_th ? 1.f : 0.f);
        float r_pen = reach_dist > far_th ? -1.f : 0.f;
        bool done = reach_dist > far_th;
        rw_done = done; rw_solved = reach_dist < near_th ? 1.f : 0.f;
        rw_dense = t.w_pose * r_reach + t.w_bonus * r_bonus + t.w_act_reg * (-act_mag) + t.w_penalty * r_pen;
        if (t.rwd && !obs_only) {
          float* r = t.rwd + (size_t)e * MM_RWD_COUNT;
          r[MM_RWD_POSE] = r_reach; r[MM_RWD_BONUS] = r_bonus; r[MM_RWD_PENALTY] = r_pen; r[MM_RWD_ACT_REG] = -act_mag;
          r[MM_RWD_SPARSE] = -reach_dist; r[MM_RWD_SOLVED] = reach_dist < near_th ? 1.f : 0.f;
          r[MM_RWD_DONE] = done ? 1.f : 0.f;
          r[MM_RWD_DENSE] = t.w_pose * r_reach + t.w_bonus * r_bonus + t.w_act_reg * (-act_mag) + t.w_penalty * r_pen;
        }
        if (t.done && !obs_only) t.done[e] = done ? 1 : 0;
      }
    }
    if (t.task == MM_TASK_WALK) {
      // obs / reward of WalkEnvV0 (walk_v0.py:283-325, 367-540); self.steps == step_count BEFORE this step's increment
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      const int nq2 = d.nq - 2;
      const int o_qv = nq2, o_cv = o_qv + d.nv, o_tq = o_cv + 2, o_fh = o_tq + 4, o_h = o_fh + 2, o_fr = o_h + 1,
                o_ph = o_fr + 6, o_ml = o_ph + 1, o_mv = o_ml + d.nu, o_mf = o_mv + d.nu, o_act = o_mf + d.nu;
      const bool isb = g > 0 && g < d.nbody;
      float ms = isb ? MF_(BODY_MASS)[g] : 0.f;
      if (a.s.body_mass_env && g == a.s.body_mass_env_id) ms = a.s.body_mass_env[e];
      const float mtot = gsum<G>(ms);
      // com velocity with the reference's sign convention: mean of -cvel[:, 3:5]
      const float cvx = gsum<G>(ms * -E.b_cvel[3]) / mtot, cvy = gsum<G>(ms * -E.b_cvel[4]) / mtot;
      const float height = gsum<G>(ms * E.b_xipos.z) / mtot + d.oz;
      const int bp = t.walk_body[0], bt = t.walk_body[1], bl = t.walk_body[2], br = t.walk_body[3];
      const V3 xp = v3(bc<G>(E.b_xpos.x, bp), bc<G>(E.b_xpos.y, bp), bc<G>(E.b_xpos.z, bp));
      const V3 xl = v3(bc<G>(E.b_xpos.x, bl), bc<G>(E.b_xpos.y, bl), bc<G>(E.b_xpos.z, bl));
      const V3 xr = v3(bc<G>(E.b_xpos.x, br), bc<G>(E.b_xpos.y, br), bc<G>(E.b_xpos.z, br));
      const float tq0 = bc<G>(E.b_xquat.w, bt), tq1 = bc<G>(E.b_xquat.x, bt), tq2 = bc<G>(E.b_xquat.y, bt), tq3 = bc<G>(E.b_xquat.z, bt);
      const float phase = fmodf((float)sc0 / (float)t.walk_hip_period, 1.f);
      float act2 = 0.f;
      if (ob) {
        for (int i = g; i < nq2; i += G) ob[i] = W[L.qpos + 2 + i];
        if (g < d.nv) ob[o_qv + g] = E.d_qvel * t.obs_dt;
      }
      for (int i = g; i < d.nu; i += G) {
        if (ob) {
          ob[o_ml + i] = W[L.actlen + i];
          ob[o_mv + i] = clampf(W[L.actvel + i], -100.f, 100.f);
          ob[o_mf + i] = clampf(W[L.actfrc + i] / 1000.f, -100.f, 100.f);
        }
      }
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[o_act + i] = x;
      }
      act2 = gsum<G>(act2);
      if (g == 0) {
        if (ob) {
          ob[o_cv] = cvx; ob[o_cv + 1] = cvy;
          ob[o_tq] = tq0; ob[o_tq + 1] = tq1; ob[o_tq + 2] = tq2; ob[o_tq + 3] = tq3;
          ob[o_fh] = xl.z + d.oz; ob[o_fh + 1] = xr.z + d.oz;
          ob[o_h] = height;
          st3(ob + o_fr, xl - xp); st3(ob + o_fr + 3, xr - xp);
          ob[o_ph] = phase;
        }
        const float* q = W + L.qpos;
        const float dvy = t.walk_target_y_vel - cvy, dvx = t.walk_target_x_vel - cvx;
        const float vel_reward = expf(-dvy * dvy) + expf(-dvx * dvx);
        const float two_pi = 6.283185307179586f;
        const float des_l = 0.8f * cosf(phase * two_pi + 3.141592653589793f), des_r = 0.8f * cosf(phase * two_pi);
        const float el = des_l - q[t.walk_qadr[0]], er = des_r - q[t.walk_qadr[1]];
        const float cyclic_hip = sqrtf(el * el + er * er);
        float rr = 0.f;
        for (int k = 0; k < 4; k++) { float dq = 5.f * (q[3 + k] - t.walk_target_rot[k]); rr += dq * dq; }
        const float ref_rot = expf(-sqrtf(rr));
        const float mag = 0.25f * (fabsf(q[t.walk_qadr[2]]) + fabsf(q[t.walk_qadr[3]]) + fabsf(q[t.walk_qadr[4]]) + fabsf(q[t.walk_qadr[5]]));
        const float joint_angle_rew = expf(-5.f * mag);
        const float act_mag = d.na != 0 ? sqrtf(act2) / (float)d.na : 0.f;
        // |(quat2mat(qpos[3:7]) @ [1,0,0])[0]| > max_rot   (walk_v0.py:514-526)
        const float nq_ = q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[6] * q[6];   // quat_math.py:151-174
        const float r00 = nq_ > 1.1920929e-07f * 4.f ? 1.f - (2.f / nq_) * (q[5] * q[5] + q[6] * q[6]) : 1.f;
        const bool done = height < t.walk_min_height || fabsf(r00) > t.walk_max_rot;
        rw_done = done; rw_solved = vel_reward >= 1.f ? 1.f : 0.f;
        rw_dense = t.walk_w[0] * vel_reward + t.walk_w[1] * (done ? 1.f : 0.f) + t.walk_w[2] * cyclic_hip +
                   t.walk_w[3] * ref_rot + t.walk_w[4] * joint_angle_rew;
        if (t.rwd && !obs_only) {   // the reset observation leaves the terminal step's reward terms in place
          float* r = t.rwd + (size_t)e * MM_RWDW_COUNT;
          r[MM_RWDW_VEL] = vel_reward; r[MM_RWDW_CYCLIC_HIP] = cyclic_hip; r[MM_RWDW_REF_ROT] = ref_rot;
          r[MM_RWDW_JOINT_ANGLE] = joint_angle_rew; r[MM_RWDW_ACT_MAG] = act_mag; r[MM_RWDW_SPARSE] = vel_reward;
          r[MM_RWDW_SOLVED] = vel_reward >= 1.f ? 1.f : 0.f; r[MM_RWDW_DONE] = done ? 1.f : 0.f;
          r[MM_RWDW_DENSE] = t.walk_w[0] * vel_reward + t.walk_w[1] * (done ? 1.f : 0.f) + t.walk_w[2] * cyclic_hip +
                             t.walk_w[3] * ref_rot + t.walk_w[4] * joint_angle_rew;
        }
        if (t.done && !obs_only) t.done[e] = done ? 1 : 0;
      }
    }
    if (t.task == MM_TASK_OBJHOLD) {
      // obs / reward of ObjHoldFixedEnvV0 (obj_hold_v0.py:82-131)
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      const int nh = d.nq - 7, nhv = d.nv - 6;
      float act2 = 0.f;
      if (ob) {
        for (int i = g; i < nh; i += G) ob[i] = W[L.qpos + i];
        if (g < nhv) ob[nh + g] = E.d_qvel * t.obs_dt;
      }
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[nh + nhv + 6 + i] = x;
      }
      act2 = gsum<G>(act2);
      if (g == 0) {
        const V3 op_i = E.site_pos(t.tip_sites[0]);
        const V3 op = op_i + E.origin();
        const V3 er = (ld3(t.target_pos + (size_t)e * 3) - E.origin()) - op_i;
        if (ob) { st3(ob + nh + nhv, op); st3(ob + nh + nhv + 3, er); }
        const float goal_dist = sqrtf(dot(er, er)), act_mag = d.na != 0 ? sqrtf(act2) / (float)d.na : 0.f;
        const float goal_th = 0.010f;
        const bool drop = goal_dist > 0.300f;
        const float bonus = (goal_dist < 2.f * goal_th ? 1.f : 0.f) + (goal_dist < goal_th ? 1.f : 0.f);
        rw_done = drop; rw_solved = goal_dist < goal_th ? 1.f : 0.f;
        rw_dense = t.w_pose * -goal_dist + t.w_bonus * bonus + t.w_act_reg * -act_mag + t.w_penalty * (drop ? -1.f : 0.f);
        if (t.rwd && !obs_only) {
          float* r = t.rwd + (size_t)e * MM_RWD_COUNT;
          r[MM_RWD_POSE] = -goal_dist; r[MM_RWD_BONUS] = bonus; r[MM_RWD_PENALTY] = drop ? -1.f : 0.f; r[MM_RWD_ACT_REG] = -act_mag;
          r[MM_RWD_SPARSE] = -goal_dist; r[MM_RWD_SOLVED] = goal_dist < goal_th ? 1.f : 0.f; r[MM_RWD_DONE] = drop ? 1.f : 0.f;
          r[MM_RWD_DENSE] = t.w_pose * -goal_dist + t.w_bonus * bonus + t.w_act_reg * -act_mag + t.w_penalty * (drop ? -1.f : 0.f);
        }
        if (t.done && !obs_only) t.done[e] = drop ? 1 : 0;
      }
    }
    if (t.task == MM_TASK_KEYTURN) {
      // obs / reward of KeyTurnEnvV0 (key_turn_v0.py:101-150)
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      const int nh = d.nq - 1, nhv = d.nv - 1;
      const int o_kq = nh + nhv, o_if = o_kq + 2, o_th = o_if + 3, o_act = o_th + 3;
      float act2 = 0.f;
      if (ob) {
        for (int i = g; i < nh; i += G) ob[i] = W[L.qpos + i];
        if (g < nhv) ob[nh + g] = E.d_qvel * t.obs_dt;
        if (g == nhv) ob[o_kq + 1] = E.d_qvel * t.obs_dt;
      }
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[o_act + i] = x;
      }
      act2 = gsum<G>(act2);
      if (g == 0) {
        const V3 kh = E.site_pos(t.tip_sites[0]);
        const V3 ifa = kh - E.site_pos(t.tip_sites[1]), tha = kh - E.site_pos(t.tip_sites[2]);
        const float key_pos = W[L.qpos + nh];
        if (ob) { ob[o_kq] = key_pos; st3(ob + o_if, ifa); st3(ob + o_th, tha); }
        const float ifd = fabsf(sqrtf(dot(ifa, ifa)) - 0.030f), thd = fabsf(sqrtf(dot(tha, tha)) - 0.030f);
        const float act_mag = d.na != 0 ? sqrtf(act2) / (float)d.na : 0.f;
        const float far_th = 0.1f, pi_ = 3.14159265358979f;
        const float bonus = (key_pos > 0.5f * pi_ ? 1.f : 0.f) + (key_pos > pi_ ? 1.f : 0.f);
        const float penalty = -(ifd > 0.5f * far_th ? 1.f : 0.f) - (thd > 0.5f * far_th ? 1.f : 0.f);
        const bool done = ifd > far_th || thd > far_th;
        rw_done = done; rw_solved = key_pos > t.key_goal_th ? 1.f : 0.f;
        rw_dense = t.key_w[0] * key_pos + t.key_w[1] * -ifd + t.key_w[2] * -thd + t.key_w[3] * -act_mag +
                   t.key_w[4] * bonus + t.key_w[5] * penalty;
        if (t.rwd && !obs_only) {
          float* r = t.rwd + (size_t)e * MM_RWDK_COUNT;
          r[MM_RWDK_KEY_TURN] = key_pos; r[MM_RWDK_IF_APPROACH] = -ifd; r[MM_RWDK_TH_APPROACH] = -thd; r[MM_RWDK_ACT_REG] = -act_mag;
          r[MM_RWDK_BONUS] = bonus; r[MM_RWDK_PENALTY] = penalty; r[MM_RWDK_SPARSE] = key_pos;
          r[MM_RWDK_SOLVED] = key_pos > t.key_goal_th ? 1.f : 0.f; r[MM_RWDK_DONE] = done ? 1.f : 0.f;
          r[MM_RWDK_DENSE] = t.key_w[0] * key_pos + t.key_w[1] * -ifd + t.key_w[2] * -thd + t.key_w[3] * -act_mag +
                             t.key_w[4] * bonus + t.key_w[5] * penalty;
        }
        if (t.done && !obs_only) t.done[e] = done ? 1 : 0;
      }
    }
    if (t.task == MM_TASK_REORIENT) {
      // obs / reward of ProprioceptiveEnvV0 (reorient_sar_v0.py:116-174)
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      const int nh = d.nq - 6;
      const int o_pos = nh, o_vel = o_pos + 3, o_rot = o_vel + 6, o_des = o_rot + 3, o_ep = o_des + 3, o_er = o_ep + 3,
                o_ml = o_er + 3, o_mv = o_ml + d.nu, o_mf = o_mv + d.nu, o_act = t.reor_obs_muscle ? o_mf + d.nu : o_ml;
      float act2 = 0.f;
      if (ob) {
        for (int i = g; i < nh; i += G) ob[i] = W[L.qpos + i];
        if (g < d.nv && g >= d.nv - 6) ob[o_vel + g - (d.nv - 6)] = E.d_qvel * t.obs_dt;
        if (t.reor_obs_muscle)
          for (int i = g; i < d.nu; i += G) { ob[o_ml + i] = W[L.actlen + i]; ob[o_mv + i] = W[L.actvel + i]; ob[o_mf + i] = W[L.actfrc + i]; }
      }
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[o_act + i] = x;
      }
      act2 = gsum<G>(act2);
      if (g == 0) {
        const int bo = t.reor_obj_body;
        const V3 opos_i = ld3(W + L.xpos + 3 * bo);
        const V3 opos = opos_i + E.origin();
        const float* R = W + L.xmat + 9 * bo;
        const float sc_ = 2.f * t.reor_axis_half[e] / t.reor_pen_length;   // pen_v0.py: the same vector through the top / bottom sites
        V3 orot = v3(R[2] * sc_, R[5] * sc_, R[8] * sc_);
        V3 odes = ld3(t.reor_des_rot + (size_t)e * 3);
        V3 epos = opos_i - E.site_pos(t.reor_eps_site), erot = orot - odes;
        if (ob) { st3(ob + o_pos, opos); st3(ob + o_rot, orot); st3(ob + o_des, odes); st3(ob + o_ep, epos); st3(ob + o_er, erot); }
        const float pos_align = sqrtf(dot(epos, epos));
        float nrm = sqrtf(dot(orot, orot)) * sqrtf(dot(odes, odes));
        if (nrm == 0.f) nrm = 1.f;                                   // vector_math.py:26-32
        const float rot_align = dot(orot, odes) / nrm;
        const bool dropped = pos_align > 0.075f;
        const float act_mag = d.na != 0 ? sqrtf(act2) / (float)d.na : 0.f;
        const float bonus = ((rot_align > 0.9f && pos_align < 0.075f) ? 1.f : 0.f) + ((rot_align > 0.95f && pos_align < 0.075f) ? 5.f : 0.f);
        rw_done = dropped; rw_solved = (rot_align > 0.95f && !dropped) ? 1.f : 0.f;
        rw_dense = t.reor_w[0] * -pos_align + t.reor_w[1] * rot_align + t.reor_w[2] * -act_mag +
                   t.reor_w[3] * (dropped ? -1.f : 0.f) + t.reor_w[4] * bonus;
        if (t.rwd && !obs_only) {
          float* r = t.rwd + (size_t)e * MM_RWDR_COUNT;
          r[MM_RWDR_POS_ALIGN] = -pos_align; r[MM_RWDR_ROT_ALIGN] = rot_align; r[MM_RWDR_ACT_REG] = -act_mag;
          r[MM_RWDR_DROP] = dropped ? -1.f : 0.f; r[MM_RWDR_BONUS] = bonus; r[MM_RWDR_SPARSE] = -pos_align + rot_align;
          r[MM_RWDR_SOLVED] = (rot_align > 0.95f && !dropped) ? 1.f : 0.f; r[MM_RWDR_DONE] = dropped ? 1.f : 0.f;
          r[MM_RWDR_DENSE] = t.reor_w[0] * -pos_align + t.reor_w[1] * rot_align + t.reor_w[2] * -act_mag +
                             t.reor_w[3] * (dropped ? -1.f : 0.f) + t.reor_w[4] * bonus;
        }
        if (t.done && !obs_only) t.done[e] = dropped ? 1 : 0;
      }
    }
    if constexpr (FOLD) {
      if (pass == 0 && has_ro && !obs_only && KA().ro.autoreset && (t.task == MM_TASK_WALK || t.task == MM_TASK_REORIENT)) {
        const bool trunc_ = t.max_episode_steps > 0 && sc >= t.max_episode_steps;
        refold = __builtin_amdgcn_readfirstlane((int)(rw_done || trunc_)) != 0;     // rw_done is lane 0's
      }
    }
    if (g == 0 && !obs_only) {
      const bool trunc = t.max_episode_steps > 0 && sc >= t.max_episode_steps;
      if (t.step_count) t.step_count[e] = (will_reset || refold) ? 0 : sc;
      if (t.truncated) t.truncated[e] = trunc ? 1 : 0;
      if (has_ro) {   // rollout bookkeeping (mm_rollout): what mm_episode_stats does in its own launch
        const __attribute__((address_space(4))) mm_rollout& ro = KA().ro;
        if (ro.ep_stats) {
          float* st = ro.ep_stats + (size_t)e * 3;
          st[0] += rw_dense; st[1] += 1.f; st[2] = fmaxf(st[2], rw_solved);
        }
        if (ro.reset_mask) ro.reset_mask[e] = (rw_done || trunc) ? 1 : 0;
      }
    }
  }
  };   // stage

  if (!dup) {   // surplus groups never write
  stage(obs_only, 0);
  if constexpr (FOLD) {
    if (refold) {
      // ---- re-arm this env inside the launch (mm_rollout.autoreset; WALK: mm_walk_reset, REORIENT: mm_reorient_reset_typed --
      // same Philox counters, keyed by the global env index and the episode counter), then run the reset-observation pass.  The
      // terminal step's reward / done / statistics are already written; its observation row is replaced by the new episode's.
      const __attribute__((address_space(4))) mm_rollout& ro = KA().ro;
      if constexpr (Engine<G, NVP, GEN, INTEG>::TW) {
        // two-wave launches: nobody waited for the helper's last job of the final forward pass (Euler's factor / the W matrix in
        // the second tile), which the next forward pass rewrites
        if (two_wave && (Engine<G, NVP, GEN, INTEG>::IMPL || (!Engine<G, NVP, GEN, INTEG>::SP && d.any_damping && d.eulerdamp))) E.tw_wait(3, E.tw_n);
      }
      const int ep = ro.episode[e];
      const uint32_t ge = (uint32_t)(a.s.env_index_base + e);
      const uint64_t sd = ro.reset_seed;
      if (t.task == MM_TASK_WALK) {
        const float *kq = ro.walk_ka_qpos, *kv = ro.walk_ka_qvel;
        if (ro.walk_random) {      // walk_v0.py:327-352: coin between the two stride keys, N(0, 0.02) on every coordinate but root height / quaternion
          uint32_t c[4] = {0xFFFFu, 2u, ge, (uint32_t)ep};
          philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
          if (!(u01(c[0]) < 0.5f)) { kq = ro.walk_kb_qpos; kv = ro.walk_kb_qvel; }
        }
        for (int i = g; i < d.nq; i += G) {
          float q = kq[i];
          if (ro.walk_random && !(i >= 2 && i < 7)) {
            uint32_t c[4] = {(uint32_t)i, 2u, ge, (uint32_t)ep};
            philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
            const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = u01(c[1]);
            q += 0.02f * sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
          }
          W[L.qpos + i] = q;
        }
        if (g < d.nv) { E.d_qvel = kv[g]; W[L.qvel + g] = E.d_qvel; }
      } else {                     // MM_TASK_REORIENT (reorient_sar_v0.py:386-432)
        uint32_t c[4] = {0u, 3u, ge, (uint32_t)ep};
        philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
        int idx = (int)(u01(c[0]) * (float)ro.reor_ntab);
        if (idx >= ro.reor_ntab) idx = ro.reor_ntab - 1;
        int ty = (int)(u01(c[3]) * 4.f);
        if (ty > 3) ty = 3;
        const float* sz = ro.reor_size_tables + 3 * (ty * ro.reor_ntab + idx);
        const float s0 = sz[0], s1 = sz[1], s2 = sz[2];
        const float ah = ty == 0 ? 1.3f * s1 : (ty == 2 ? s1 : s2);
        const float e0 = -1.f + 2.f * u01(c[1]), e1 = -0.8f + 2.f * u01(c[2]);
        const float aj = -0.5f * e1, ak = 0.5f * e0;
        const float sj = sinf(aj), cj = cosf(aj), sk = sinf(ak), ck = cosf(ak);
        const float qw = cj * ck, qx = cj * sk, qy = -(sj * ck), qz = -sj * sk;
        const float sc_ = 2.f * ah / ro.reor_tar_length;
        if (g == 0) {             // every lane computed the same draws; lane 0 publishes the per-env model deltas
          ro.reor_geom_type_env[e] = MM_GEOM_CAPSULE + ty;
          ro.reor_geom_size_env[(size_t)e * 3] = s0; ro.reor_geom_size_env[(size_t)e * 3 + 1] = s1; ro.reor_geom_size_env[(size_t)e * 3 + 2] = s2;
          ro.reor_axis_half[e] = ah;
          ro.reor_des_rot[(size_t)e * 3 + 0] = 2.f * (qx * qz + qw * qy) * sc_;
          ro.reor_des_rot[(size_t)e * 3 + 1] = 2.f * (qy * qz - qw * qx) * sc_;
          ro.reor_des_rot[(size_t)e * 3 + 2] = (1.f - 2.f * (qx * qx + qy * qy)) * sc_;
        }
        E.env_gtype = MM_GEOM_CAPSULE + ty; E.env_has_gs = true; E.env_gsv[0] = s0; E.env_gsv[1] = s1; E.env_gsv[2] = s2;
        for (int i = g; i < d.nq; i += G) W[L.qpos + i] = ro.reor_init_qpos[i];
        if (g < d.nv) { E.d_qvel = 0.f; W[L.qvel + g] = 0.f; }
      }
      if (g < d.nv) E.d_warm = 0.f;
      for (int i = g; i < d.na; i += G) W[L.act + i] = 0.f;
      if (t.fatigue)               // CumulativeFatigue.reset (fatigue.py:82-99): MF = fatigue_reset_vec (or 0), MR = 1 - MF, MA = 0
        for (int i = g; i < d.na; i += G) {
          const float mf = ro.fat_reset_vec ? ro.fat_reset_vec[i] : 0.f;
          const size_t k = (size_t)e * d.na + i;
          t.fat_MA[k] = 0.f; t.fat_MR[k] = 1.f - mf; t.fat_MF[k] = mf;
        }
      time = 0.f; E.status = 0;
      if (g == 0) ro.episode[e] = ep + 1;
      E.reinit_transients();
      GSYNC();
      E.run(0, true, time);
      stage(true, 1);
    }
  }
  // ---- store state
  if (!will_reset) {
    for (int i = g; i < d.nq; i += G) a.s.qpos[(size_t)e * d.nq + i] = W[L.qpos + i];
    if (g < d.nv) {
      a.s.qvel[(size_t)e * d.nv + g] = E.d_qvel;
      a.s.qacc_warmstart[(size_t)e * d.nv + g] = E.d_warm;
    }
    for (int i = g; i < d.na; i += G) a.s.act[(size_t)e * d.na + i] = W[L.act + i];
    if (g == 0) { a.s.time[e] = time; if (a.s.status) a.s.status[e] = E.status; }
  } else {
    // masked auto-reset of a POSE env folded into this launch: the draws, state and first observation k_reset produces for
    // mm_pose_reset (pose_v0.py:174-257; Philox counter (i/2, 0, global env, episode): words 0/1 -> qpos, 2/3 -> target)
    const __attribute__((address_space(4))) mm_rollout& ro = KA().ro;
    const int ep = ro.episode[e];
    const uint64_t sd = ro.reset_seed;
    const int o_err = t.obs_layout == 1 ? d.nq + d.nv + d.na : d.nq + d.nv;
    const int o_act = t.obs_layout == 1 ? d.nq + d.nv : 2 * d.nq + d.nv;
    float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
    for (int i = g; i < d.nq; i += G) {
      uint32_t c[4] = {(uint32_t)(i >> 1), 0u, (uint32_t)(a.s.env_index_base + e), (uint32_t)ep};
      philox4x32_10(c, (uint32_t)sd, (uint32_t)(sd >> 32));
      const float uq = u01((i & 1) ? c[1] : c[0]), ut = u01((i & 1) ? c[3] : c[2]);
      const float q = ro.random_qpos ? ro.qlo[i] + (ro.qhi[i] - ro.qlo[i]) * uq : MF_(QPOS0)[i];
      const float tg = ro.tlo[i] + (ro.thi[i] - ro.tlo[i]) * ut;
      a.s.qpos[(size_t)e * d.nq + i] = q;
      ro.target[(size_t)e * d.nq + i] = tg;
      if (ob) { ob[i] = q; ob[o_err + i] = tg - q; }
    }
    if (g < d.nv) {
      a.s.qvel[(size_t)e * d.nv + g] = 0.f;
      a.s.qacc_warmstart[(size_t)e * d.nv + g] = 0.f;
      if (ob) ob[d.nq + g] = 0.f;
    }
    for (int i = g; i < d.na; i += G) { a.s.act[(size_t)e * d.na + i] = 0.f; if (ob) ob[o_act + i] = 0.f; }
    if (g == 0) { a.s.time[e] = 0.f; if (a.s.status) a.s.status[e] = 0; ro.episode[e] = ep + 1; }
  }
  }   // !dup
  if constexpr (Engine<G, NVP, GEN, INTEG>::TW) { if (two_wave) E.tw_signal(0, Engine<G, NVP, GEN, INTEG>::TW_DONE); }
}

