// myosim_engine.hip -- MI355X (gfx950 / CDNA4) batched musculoskeletal physics step.
//
// Execution model ("wave-cooperative"): every environment is owned by a GROUP of G
// adjacent lanes of one 64-wide wavefront (G in {4,8,16,32,64}, 64/G envs per wave).
// The env's whole mjData-like workspace lives in LDS for the duration of the fused
// env-step (frame_skip substeps + final forward + obs/reward); HBM is touched once
// to load state/action and once to store state/obs/reward.  Inside a stage the G
// lanes sweep the stage's natural index set (bodies of one tree level, tendons,
// dofs, constraint rows ...); tree recursions advance level by level.  Lanes of a
// group exchange data through LDS; a wavefront executes in lock-step and the LDS
// services one wave's instructions in order, so stage boundaries need only a
// compiler fence (GSYNC), never s_barrier.  Reductions use cross-lane shuffles.
//
// Pipeline restated (stage order of mj_step; reference call
// site myosuite/robot/robot.py:856-861):  kinematics -> comPos -> tendon(+wrap) ->
// comVel/RNE -> CRB -> L'DL -> passive/actuation -> constraint rows -> Newton ->
// semi-implicit Euler (implicit joint damping).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>

#include "../../include/myosim_model.h"
#include "../../include/myosim.h"

#define MINVALF 1e-15f

// ------------------------------------------------------------------ kernel args
struct Dims {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, neq, npair, nM, nlevel, ndoflevel, njmax, ntenJ;
  int iterations, ls_iterations, eulerdamp, any_damping;
  float timestep, gx, gy, gz, tolerance, ls_tolerance, meaninertia;
};

// LDS workspace layout (offsets in 32-bit words from the env's base)
struct Layout {
  int qpos, qvel, act, ctrl, warm;
  int xpos, xquat, xmat, xipos, xanchor, xaxis, com;
  int cinert, cdof, cdofdot, cvel, cacc;
  int tenlen, tenvel, tenj, tenfrc, actlen, actvel, actfrc, actdot;
  int qM, qLD, qH, dinv, hdinv;
  int bias, passive, smooth, qaccsm, qacc, qfrccon, Ma, grad, search, Mv, tmp;
  int efc_kind, efc_id, efc_pos, efc_D, efc_aref, efc_jar, efc_jv, efc_frc;
  int total;
};

// engine-private tables appended behind the model blob on the device
struct Aux {
  int dof_ndesc, dof_depth;        // [nv]
  int dofj_adr, dofj_entry, dofj_tendon;  // transpose of the sparse tendon Jacobian
  int root_list, nroot;            // bodies that root a kinematic tree
};

struct KArgs {
  const uint32_t* blob;
  int sec[MM_NSEC];
  Dims d;
  Layout L;
  Aux x;
  mm_state s;
  const float* ctrl;     // [nenv][nu] action / control input
  mm_task t;             // task.task == MM_TASK_NONE for plain mm_step / mm_forward
  mm_derived o;
  int has_derived;
  int mode;              // 0: step(s) only, 1: forward only, 2: env step
  float* dbg;            // optional [nenv][L.total] workspace dump after the last forward
};

#define MI_(S) (reinterpret_cast<const int*>(a.blob + a.sec[MM_SEC_##S]))
#define MF_(S) (reinterpret_cast<const float*>(a.blob + a.sec[MM_SEC_##S]))
#define AUXI(f) (reinterpret_cast<const int*>(a.blob + a.x.f))

// stage boundary inside one wavefront: LDS traffic of a wave is serviced in program
// order, so only the compiler must be kept from moving LDS accesses across.
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
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
struct Q4 { float w, x, y, z; };
__device__ __forceinline__ Q4 ldq(const float* p) { Q4 q = {p[0], p[1], p[2], p[3]}; return q; }
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
__device__ __forceinline__ M3 ldm(const float* p) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
__device__ __forceinline__ V3 mv(const M3& m, V3 v) {
  return v3(m.m[0] * v.x + m.m[1] * v.y + m.m[2] * v.z, m.m[3] * v.x + m.m[4] * v.y + m.m[5] * v.z,
            m.m[6] * v.x + m.m[7] * v.y + m.m[8] * v.z);
}
__device__ __forceinline__ V3 mtv(const M3& m, V3 v) {
  return v3(m.m[0] * v.x + m.m[3] * v.y + m.m[6] * v.z, m.m[1] * v.x + m.m[4] * v.y + m.m[7] * v.z,
            m.m[2] * v.x + m.m[5] * v.y + m.m[8] * v.z);
}
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

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
  V3 a = cross(w, sa), b = cross(w, sl) + cross(l, sa);
  st3(res, a); st3(res + 3, b);
}
__device__ __forceinline__ void cross_force(float* res, const float* v, const float* f) {
  V3 w = ld3(v), l = ld3(v + 3), fa = ld3(f), fl = ld3(f + 3);
  st3(res, cross(w, fa) + cross(l, fl));
  st3(res + 3, cross(w, fl));
}

// ---------------------------------------------------------------- group helpers
template <int G>
__device__ __forceinline__ float gsum(float v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, G);
  return v;
}
template <int G>
__device__ __forceinline__ int gor(int v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v |= __shfl_xor(v, m, G);
  return v;
}

// ------------------------------------------------------------- tendon wrapping
__device__ __forceinline__ bool seg_intersect(float p1x, float p1y, float p2x, float p2y, float p3x, float p3y,
                                               float p4x, float p4y) {
  float det = (p4y - p3y) * (p2x - p1x) - (p4x - p3x) * (p2y - p1y);
  if (fabsf(det) < MINVALF) return false;
  float a = ((p4x - p3x) * (p1y - p3y) - (p4y - p3y) * (p1x - p3x)) / det;
  float b = ((p2x - p1x) * (p1y - p3y) - (p2y - p1y) * (p1x - p3x)) / det;
  return a >= 0.f && a <= 1.f && b >= 0.f && b <= 1.f;
}

// 2-D wrap around origin-centred circle; returns arc length or -1; pnt = tangent points
__device__ __forceinline__ float wrap_circle(float pnt[4], float d0x, float d0y, float d1x, float d1y, bool has_side, float sdx,
                             float sdy, float radius) {
  float sqlen0 = d0x * d0x + d0y * d0y, sqlen1 = d1x * d1x + d1y * d1y, sqrad = radius * radius;
  float difx = d1x - d0x, dify = d1y - d0y;
  float dd = difx * difx + dify * dify;
  float aa = -(difx * d0x + dify * d0y) / fmaxf(dd, MINVALF);
  aa = clampf(aa, 0.f, 1.f);
  float tx = d0x + aa * difx, ty = d0y + aa * dify;
  if (tx * tx + ty * ty > sqrad && (!has_side || sdx * tx + sdy * ty >= 0.f)) return -1.f;
  if (sqlen0 < sqrad || sqlen1 < sqrad) return -1.f;
  float sqrt0 = sqrtf(sqlen0 - sqrad), sqrt1 = sqrtf(sqlen1 - sqrad);
  float sol[2][4], good[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    float sgn = i == 0 ? 1.f : -1.f;
    sol[i][0] = (d0x * sqrad + sgn * radius * d0y * sqrt0) / sqlen0;
    sol[i][1] = (d0y * sqrad - sgn * radius * d0x * sqrt0) / sqlen0;
    sol[i][2] = (d1x * sqrad - sgn * radius * d1y * sqrt1) / sqlen1;
    sol[i][3] = (d1y * sqrad + sgn * radius * d1x * sqrt1) / sqlen1;
    if (has_side) {
      float ux = sol[i][0] + sol[i][2], uy = sol[i][1] + sol[i][3];
      float n = fmaxf(sqrtf(ux * ux + uy * uy), MINVALF);
      good[i] = (ux * sdx + uy * sdy) / n;
    } else {
      float ux = sol[i][0] - sol[i][2], uy = sol[i][1] - sol[i][3];
      good[i] = -(ux * ux + uy * uy);
    }
    if (seg_intersect(d0x, d0y, sol[i][0], sol[i][1], d1x, d1y, sol[i][2], sol[i][3])) good[i] = -10000.f;
  }
  int i = good[0] > good[1] ? 0 : 1;
  pnt[0] = sol[i][0]; pnt[1] = sol[i][1]; pnt[2] = sol[i][2]; pnt[3] = sol[i][3];
  if (seg_intersect(d0x, d0y, pnt[0], pnt[1], d1x, d1y, pnt[2], pnt[3])) return -1.f;
  float c = clampf((pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqrad, -1.f, 1.f);
  return radius * acosf(c);
}

// 3-D wrap over sphere / cylinder; w0,w1 world surface points; returns arc length or -1
__device__ __forceinline__ float wrap_geom(V3& w0, V3& w1, V3 x0, V3 x1, V3 gpos, const M3& gmat, float radius, bool is_cyl,
                           bool has_side, V3 side) {
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

// ------------------------------------------------------------------ muscle model
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

// =========================================================================== engine
// All member functions are collective over the G lanes of one env group.
template <int G>
struct Engine {
  const KArgs& a;
  float* W;   // LDS workspace of this env
  int g;      // lane within group
  int nefc;   // constraint rows of the current forward pass (group-uniform)
  int niter;  // Newton iterations of the last solve (group-uniform)
  int status; // sticky status bits (group-uniform)

  __device__ Engine(const KArgs& a_, float* W_, int g_) : a(a_), W(W_), g(g_), nefc(0), niter(0), status(0) {}

  // ---------------------------------------------------------------- A1 kinematics
  __device__ __forceinline__ void kinematics() {
    const Layout& L = a.L;
    if (g == 0) {
      st3(W + L.xpos, v3(0.f, 0.f, 0.f));
      W[L.xquat] = 1.f; W[L.xquat + 1] = 0.f; W[L.xquat + 2] = 0.f; W[L.xquat + 3] = 0.f;
      for (int k = 0; k < 9; k++) W[L.xmat + k] = (k == 0 || k == 4 || k == 8) ? 1.f : 0.f;
      st3(W + L.xipos, v3(0.f, 0.f, 0.f));
    }
    GSYNC();
    const int* lv_adr = MI_(LEVEL_ADR); const int* lv_body = MI_(LEVEL_BODY);
    const int* parent = MI_(BODY_PARENT);
    for (int lv = 0; lv < a.d.nlevel; lv++) {
      int i0 = lv_adr[lv], i1 = lv_adr[lv + 1];
      for (int idx = i0 + g; idx < i1; idx += G) {
        int b = lv_body[idx], p = parent[b];
        M3 pm = ldm(W + L.xmat + 9 * p);
        V3 pos = ld3(W + L.xpos + 3 * p) + mv(pm, ld3(MF_(BODY_POS) + 3 * b));
        Q4 quat = qmul(ldq(W + L.xquat + 4 * p), ldq(MF_(BODY_QUAT) + 4 * b));
        int ja = MI_(BODY_JNTADR)[b], jn = MI_(BODY_JNTNUM)[b];
        for (int j = ja; j < ja + jn; j++) {
          int type = MI_(JNT_TYPE)[j], qa = MI_(JNT_QPOSADR)[j];
          if (type == MM_JNT_FREE) {
            pos = ld3(W + L.qpos + qa);
            quat = qnorm(ldq(W + L.qpos + qa + 3));
            st3(W + L.xanchor + 3 * j, pos);
            M3 m = q2m(quat);
            st3(W + L.xaxis + 3 * j, v3(m.m[2], m.m[5], m.m[8]));
            continue;
          }
          M3 m = q2m(quat);
          V3 jpos = ld3(MF_(JNT_POS) + 3 * j), jax = ld3(MF_(JNT_AXIS) + 3 * j);
          V3 anchor = pos + mv(m, jpos), axis = mv(m, jax);
          st3(W + L.xanchor + 3 * j, anchor);
          st3(W + L.xaxis + 3 * j, axis);
          if (type == MM_JNT_SLIDE) {
            pos = pos + (W[L.qpos + qa] - MF_(QPOS0)[qa]) * axis;
          } else if (type == MM_JNT_HINGE) {
            float ang = W[L.qpos + qa] - MF_(QPOS0)[qa];
            float sn, cs;
            sincosf(0.5f * ang, &sn, &cs);
            Q4 ql = {cs, jax.x * sn, jax.y * sn, jax.z * sn};
            quat = qmul(quat, ql);
            pos = anchor - mv(q2m(quat), jpos);
          } else {  // ball
            quat = qmul(quat, qnorm(ldq(W + L.qpos + qa)));
            pos = anchor - mv(q2m(quat), jpos);
          }
        }
        quat = qnorm(quat);
        M3 m = q2m(quat);
        st3(W + L.xpos + 3 * b, pos);
        W[L.xquat + 4 * b] = quat.w; W[L.xquat + 4 * b + 1] = quat.x;
        W[L.xquat + 4 * b + 2] = quat.y; W[L.xquat + 4 * b + 3] = quat.z;
        for (int k = 0; k < 9; k++) W[L.xmat + 9 * b + k] = m.m[k];
        st3(W + L.xipos + 3 * b, pos + mv(m, ld3(MF_(BODY_IPOS) + 3 * b)));
      }
      GSYNC();
    }
  }

  __device__ __forceinline__ V3 site_pos(int s) const {
    int b = MI_(SITE_BODYID)[s];
    return ld3(W + a.L.xpos + 3 * b) + mv(ldm(W + a.L.xmat + 9 * b), ld3(MF_(SITE_POS) + 3 * s));
  }
  __device__ __forceinline__ V3 geom_pos(int gi) const {
    int b = MI_(GEOM_BODYID)[gi];
    return ld3(W + a.L.xpos + 3 * b) + mv(ldm(W + a.L.xmat + 9 * b), ld3(MF_(GEOM_POS) + 3 * gi));
  }
  __device__ __forceinline__ M3 geom_mat(int gi) const {
    int b = MI_(GEOM_BODYID)[gi];
    return q2m(qmul(ldq(W + a.L.xquat + 4 * b), ldq(MF_(GEOM_QUAT) + 4 * gi)));
  }

  // subtree COM of each tree root, body inertias about it, dof motion axes
  __device__ __forceinline__ void com_pos() {
    const Layout& L = a.L;
    const int* rootid = MI_(BODY_ROOTID);
    const int* roots = AUXI(root_list);
    for (int r = 0; r < a.x.nroot; r++) {
      int rb = roots[r];
      float sm = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
      for (int b = 1 + g; b < a.d.nbody; b += G)
        if (rootid[b] == rb) {
          float m = MF_(BODY_MASS)[b];
          V3 p = ld3(W + L.xipos + 3 * b);
          sm += m; sx += m * p.x; sy += m * p.y; sz += m * p.z;
        }
      sm = gsum<G>(sm); sx = gsum<G>(sx); sy = gsum<G>(sy); sz = gsum<G>(sz);
      if (g == 0) {
        V3 c = sm < MINVALF ? ld3(W + L.xipos + 3 * rb) : (1.f / sm) * v3(sx, sy, sz);
        st3(W + L.com + 3 * rb, c);
      }
    }
    GSYNC();
    for (int b = 1 + g; b < a.d.nbody; b += G) {
      V3 c = ld3(W + L.com + 3 * rootid[b]);
      M3 R = q2m(qmul(ldq(W + L.xquat + 4 * b), ldq(MF_(BODY_IQUAT) + 4 * b)));
      V3 I = ld3(MF_(BODY_INERTIA) + 3 * b);
      float ms = MF_(BODY_MASS)[b];
      V3 r = ld3(W + L.xipos + 3 * b) - c;
      float xx = 0.f, yy = 0.f, zz = 0.f, xy = 0.f, xz = 0.f, yz = 0.f;
      const float Iv[3] = {I.x, I.y, I.z};
#pragma unroll
      for (int k = 0; k < 3; k++) {
        xx += R.m[k] * Iv[k] * R.m[k]; yy += R.m[3 + k] * Iv[k] * R.m[3 + k]; zz += R.m[6 + k] * Iv[k] * R.m[6 + k];
        xy += R.m[k] * Iv[k] * R.m[3 + k]; xz += R.m[k] * Iv[k] * R.m[6 + k]; yz += R.m[3 + k] * Iv[k] * R.m[6 + k];
      }
      float r2 = dot(r, r);
      float* ci = W + L.cinert + 10 * b;
      ci[0] = xx + ms * (r2 - r.x * r.x); ci[1] = yy + ms * (r2 - r.y * r.y); ci[2] = zz + ms * (r2 - r.z * r.z);
      ci[3] = xy - ms * r.x * r.y; ci[4] = xz - ms * r.x * r.z; ci[5] = yz - ms * r.y * r.z;
      ci[6] = ms * r.x; ci[7] = ms * r.y; ci[8] = ms * r.z; ci[9] = ms;
    }
    if (g == 0) for (int k = 0; k < 10; k++) W[L.cinert + k] = 0.f;
    for (int j = g; j < a.d.njnt; j += G) {
      int b = MI_(JNT_BODYID)[j], da = MI_(JNT_DOFADR)[j], type = MI_(JNT_TYPE)[j];
      V3 off = ld3(W + L.com + 3 * rootid[b]) - ld3(W + L.xanchor + 3 * j);
      if (type == MM_JNT_HINGE) {
        V3 ax = ld3(W + L.xaxis + 3 * j);
        st3(W + L.cdof + 6 * da, ax); st3(W + L.cdof + 6 * da + 3, cross(ax, off));
      } else if (type == MM_JNT_SLIDE) {
        st3(W + L.cdof + 6 * da, v3(0.f, 0.f, 0.f)); st3(W + L.cdof + 6 * da + 3, ld3(W + L.xaxis + 3 * j));
      } else {
        int r0 = da;
        if (type == MM_JNT_FREE) {
          for (int k = 0; k < 3; k++) {
            st3(W + L.cdof + 6 * (da + k), v3(0.f, 0.f, 0.f));
            st3(W + L.cdof + 6 * (da + k) + 3, v3(k == 0 ? 1.f : 0.f, k == 1 ? 1.f : 0.f, k == 2 ? 1.f : 0.f));
          }
          r0 = da + 3;
        }
        M3 R = ldm(W + L.xmat + 9 * b);
        for (int k = 0; k < 3; k++) {
          V3 ax = v3(R.m[k], R.m[3 + k], R.m[6 + k]);
          st3(W + L.cdof + 6 * (r0 + k), ax); st3(W + L.cdof + 6 * (r0 + k) + 3, cross(ax, off));
        }
      }
    }
    GSYNC();
  }

  // ---------------------------------------------------------------- A2 tendons
  // add +/- u . (translational Jacobian column) for all dofs of `body` (one body only)
  __device__ __forceinline__ void tenj_add_body(int t, int body, V3 pnt, V3 u, float sgn) {
    const Layout& L = a.L;
    int da = MI_(BODY_DOFADR)[body], dn = MI_(BODY_DOFNUM)[body];
    if (dn <= 0) return;
    V3 off = pnt - ld3(W + L.com + 3 * MI_(BODY_ROOTID)[body]);
    int j0 = MI_(TENJ_ADR)[t], j1 = MI_(TENJ_ADR)[t + 1];
    const int* tdof = MI_(TENJ_DOF);
    for (int i = da; i < da + dn; i++) {
      V3 ang = ld3(W + L.cdof + 6 * i), lin = ld3(W + L.cdof + 6 * i + 3);
      float val = sgn * dot(u, lin + cross(ang, off));
      for (int e = j0; e < j1; e++)
        if (tdof[e] == i) { W[L.tenj + e] += val; break; }
    }
  }
  // segment p0 (body b0) -> p1 (body b1), unit direction u, path divisor
  __device__ __forceinline__ void tenj_segment(int t, int b0, V3 p0, int b1, V3 p1, V3 u, float inv_div) {
    const int* parent = MI_(BODY_PARENT);
    u = inv_div * u;
    while (b0 != b1) {
      if (b0 > b1) { tenj_add_body(t, b0, p0, u, -1.f); b0 = parent[b0]; }
      else { tenj_add_body(t, b1, p1, u, 1.f); b1 = parent[b1]; }
    }
  }

  __device__ __forceinline__ void tendon() {
    const Layout& L = a.L;
    const int *wt = MI_(WRAP_TYPE), *wo = MI_(WRAP_OBJID);
    const float* wp = MF_(WRAP_PRM);
    for (int e = g; e < a.d.ntenJ; e += G) W[L.tenj + e] = 0.f;
    GSYNC();
    for (int t = g; t < a.d.ntendon; t += G) {
      int adr = MI_(TENDON_ADR)[t], num = MI_(TENDON_NUM)[t];
      float len = 0.f, inv_div = 1.f;
      for (int k = 0; k < num; k++)
        if (wt[adr + k] == MM_WRAP_JOINT) {
          int jn = wo[adr + k];
          len += wp[adr + k] * W[L.qpos + MI_(JNT_QPOSADR)[jn]];
          int dof = MI_(JNT_DOFADR)[jn];
          for (int e = MI_(TENJ_ADR)[t]; e < MI_(TENJ_ADR)[t + 1]; e++)
            if (MI_(TENJ_DOF)[e] == dof) { W[L.tenj + e] += wp[adr + k]; break; }
        }
      int j = 0;
      while (j < num - 1) {
        int t0 = wt[adr + j], t1 = wt[adr + j + 1];
        if (t0 == MM_WRAP_JOINT) { j++; continue; }
        if (t0 == MM_WRAP_PULLEY || t1 == MM_WRAP_PULLEY) {
          if (t0 == MM_WRAP_PULLEY) inv_div = 1.f / wp[adr + j];
          j++;
          continue;
        }
        int s0 = wo[adr + j];
        V3 p0 = site_pos(s0);
        int b0 = MI_(SITE_BODYID)[s0];
        if (t1 == MM_WRAP_SITE) {
          int s1 = wo[adr + j + 1];
          V3 p1 = site_pos(s1);
          int b1 = MI_(SITE_BODYID)[s1];
          V3 dif = p1 - p0;
          float n = sqrtf(dot(dif, dif));
          len += n * inv_div;
          if (b0 != b1) {
            V3 u = n < MINVALF ? v3(1.f, 0.f, 0.f) : (1.f / n) * dif;
            tenj_segment(t, b0, p0, b1, p1, u, inv_div);
          }
          j += 1;
        } else {
          int gi = wo[adr + j + 1], s1 = wo[adr + j + 2];
          V3 p1 = site_pos(s1);
          int b1 = MI_(SITE_BODYID)[s1];
          int sideid = (int)lrintf(wp[adr + j + 1]);
          V3 side = v3(0.f, 0.f, 0.f);
          if (sideid >= 0) side = site_pos(sideid);
          V3 w0, w1;
          float wlen = wrap_geom(w0, w1, p0, p1, geom_pos(gi), geom_mat(gi), MF_(GEOM_SIZE)[3 * gi],
                                 t1 == MM_WRAP_CYLINDER, sideid >= 0, side);
          if (wlen < 0.f) {
            V3 dif = p1 - p0;
            float n = sqrtf(dot(dif, dif));
            len += n * inv_div;
            if (b0 != b1) {
              V3 u = n < MINVALF ? v3(1.f, 0.f, 0.f) : (1.f / n) * dif;
              tenj_segment(t, b0, p0, b1, p1, u, inv_div);
            }
          } else {
            int bg = MI_(GEOM_BODYID)[gi];
            V3 d0 = w0 - p0, d1 = p1 - w1;
            float n0 = sqrtf(dot(d0, d0)), n1 = sqrtf(dot(d1, d1));
            len += (n0 + wlen + n1) * inv_div;
            if (b0 != bg) tenj_segment(t, b0, p0, bg, w0, n0 < MINVALF ? v3(1.f, 0.f, 0.f) : (1.f / n0) * d0, inv_div);
            if (bg != b1) tenj_segment(t, bg, w1, b1, p1, n1 < MINVALF ? v3(1.f, 0.f, 0.f) : (1.f / n1) * d1, inv_div);
          }
          j += 2;
        }
      }
      W[L.tenlen + t] = len;
    }
    GSYNC();
  }

  // ----------------------------------------------------- A5 velocity stage + bias
  __device__ __forceinline__ void velocity_bias() {
    const Layout& L = a.L;
    // tendon and actuator velocities
    for (int t = g; t < a.d.ntendon; t += G) {
      float s = 0.f;
      for (int e = MI_(TENJ_ADR)[t]; e < MI_(TENJ_ADR)[t + 1]; e++) s += W[L.tenj + e] * W[L.qvel + MI_(TENJ_DOF)[e]];
      W[L.tenvel + t] = s;
    }
    if (g == 0) {
      for (int k = 0; k < 6; k++) W[L.cvel + k] = 0.f;
      W[L.cacc] = 0.f; W[L.cacc + 1] = 0.f; W[L.cacc + 2] = 0.f;
      W[L.cacc + 3] = -a.d.gx; W[L.cacc + 4] = -a.d.gy; W[L.cacc + 5] = -a.d.gz;
    }
    GSYNC();
    const int* lv_adr = MI_(LEVEL_ADR); const int* lv_body = MI_(LEVEL_BODY);
    const int* parent = MI_(BODY_PARENT);
    // forward pass: cvel, cdof_dot, cacc, then cfrc_body (stored over cacc)
    for (int lv = 0; lv < a.d.nlevel; lv++) {
      for (int idx = lv_adr[lv] + g; idx < lv_adr[lv + 1]; idx += G) {
        int b = lv_body[idx], p = parent[b];
        float cv[6], ca[6];
        for (int k = 0; k < 6; k++) { cv[k] = W[L.cvel + 6 * p + k]; ca[k] = W[L.cacc + 6 * p + k]; }
        int ja = MI_(BODY_JNTADR)[b], jn = MI_(BODY_JNTNUM)[b];
        for (int j = ja; j < ja + jn; j++) {
          int type = MI_(JNT_TYPE)[j], da = MI_(JNT_DOFADR)[j];
          if (type == MM_JNT_FREE) {
            for (int d3 = 0; d3 < 3; d3++) {
              for (int k = 0; k < 6; k++) W[L.cdofdot + 6 * (da + d3) + k] = 0.f;
              float qv = W[L.qvel + da + d3];
              for (int k = 0; k < 6; k++) cv[k] += W[L.cdof + 6 * (da + d3) + k] * qv;
            }
            da += 3;
            type = MM_JNT_BALL;
          }
          int nd = type == MM_JNT_BALL ? 3 : 1;
          float cd[3][6], cdd[3][6];
          for (int d3 = 0; d3 < nd; d3++) {
            for (int k = 0; k < 6; k++) cd[d3][k] = W[L.cdof + 6 * (da + d3) + k];
            cross_motion(cdd[d3], cv, cd[d3]);
            for (int k = 0; k < 6; k++) W[L.cdofdot + 6 * (da + d3) + k] = cdd[d3][k];
          }
          for (int d3 = 0; d3 < nd; d3++) {
            float qv = W[L.qvel + da + d3];
            for (int k = 0; k < 6; k++) { cv[k] += cd[d3][k] * qv; ca[k] += cdd[d3][k] * qv; }
          }
        }
        for (int k = 0; k < 6; k++) { W[L.cvel + 6 * b + k] = cv[k]; W[L.cacc + 6 * b + k] = ca[k]; }
      }
      GSYNC();
    }
    // cfrc_body = I*cacc + cvel x* (I*cvel), in place over cacc (children only read parents' cacc above)
    for (int b = 1 + g; b < a.d.nbody; b += G) {
      float I[10], cv[6], ca[6], Ia[6], Iv[6], x[6];
      for (int k = 0; k < 10; k++) I[k] = W[L.cinert + 10 * b + k];
      for (int k = 0; k < 6; k++) { cv[k] = W[L.cvel + 6 * b + k]; ca[k] = W[L.cacc + 6 * b + k]; }
      inert_mul(Ia, I, ca); inert_mul(Iv, I, cv); cross_force(x, cv, Iv);
      for (int k = 0; k < 6; k++) W[L.cacc + 6 * b + k] = Ia[k] + x[k];
    }
    if (g == 0) for (int k = 0; k < 6; k++) W[L.cacc + k] = 0.f;
    GSYNC();
    // backward accumulation (children -> parent), level by level, gather form
    for (int lv = a.d.nlevel - 1; lv >= 1; lv--) {
      for (int idx = lv_adr[lv - 1] + g; idx < lv_adr[lv]; idx += G) {
        int p = lv_body[idx];
        float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        bool any = false;
        for (int c = lv_adr[lv]; c < lv_adr[lv + 1]; c++) {
          int b = lv_body[c];
          if (parent[b] == p) { any = true; for (int k = 0; k < 6; k++) acc[k] += W[L.cacc + 6 * b + k]; }
        }
        if (any) for (int k = 0; k < 6; k++) W[L.cacc + 6 * p + k] += acc[k];
      }
      GSYNC();
    }
    for (int i = g; i < a.d.nv; i += G) {
      int b = MI_(DOF_BODYID)[i];
      float s = 0.f;
      for (int k = 0; k < 6; k++) s += W[L.cdof + 6 * i + k] * W[L.cacc + 6 * b + k];
      W[L.bias + i] = s;
    }
    GSYNC();
  }

  // ---------------------------------------------------------------- A4 CRB + factor
  __device__ __forceinline__ void crb() {
    const Layout& L = a.L;
    const int* lv_adr = MI_(LEVEL_ADR); const int* lv_body = MI_(LEVEL_BODY);
    const int* parent = MI_(BODY_PARENT);
    // composite inertias accumulate IN PLACE over cinert (velocity stage already consumed it)
    for (int lv = a.d.nlevel - 1; lv >= 1; lv--) {
      for (int idx = lv_adr[lv - 1] + g; idx < lv_adr[lv]; idx += G) {
        int p = lv_body[idx];
        float acc[10];
        for (int k = 0; k < 10; k++) acc[k] = 0.f;
        bool any = false;
        for (int c = lv_adr[lv]; c < lv_adr[lv + 1]; c++) {
          int b = lv_body[c];
          if (parent[b] == p) { any = true; for (int k = 0; k < 10; k++) acc[k] += W[L.cinert + 10 * b + k]; }
        }
        if (any) for (int k = 0; k < 10; k++) W[L.cinert + 10 * p + k] += acc[k];
      }
      GSYNC();
    }
    const int *dpar = MI_(DOF_PARENTID), *madr = MI_(DOF_MADR);
    for (int i = g; i < a.d.nv; i += G) {
      float I[10], cd[6], buf[6];
      int b = MI_(DOF_BODYID)[i];
      for (int k = 0; k < 10; k++) I[k] = W[L.cinert + 10 * b + k];
      for (int k = 0; k < 6; k++) cd[k] = W[L.cdof + 6 * i + k];
      inert_mul(buf, I, cd);
      int adr = madr[i], j = i;
      bool first = true;
      while (j >= 0) {
        float s = 0.f;
        for (int k = 0; k < 6; k++) s += W[L.cdof + 6 * j + k] * buf[k];
        if (first) { s += MF_(DOF_ARMATURE)[i]; first = false; }
        W[L.qM + adr] = s;
        adr++;
        j = dpar[j];
      }
    }
    GSYNC();
  }

  // in-place sparse L'DL of the matrix stored at W[off..] (M layout); dinv at W[doff..]
  __device__ __forceinline__ void factor(int off, int doff) {
    const int *madr = MI_(DOF_MADR), *ndesc = AUXI(dof_ndesc), *depth = AUXI(dof_depth);
    const int *dl_adr = MI_(DOF_LEVEL_ADR), *dl_dof = MI_(DOF_LEVEL_DOF);
    for (int lv = a.d.ndoflevel - 1; lv >= 0; lv--) {
      for (int idx = dl_adr[lv] + g; idx < dl_adr[lv + 1]; idx += G) {
        int i = dl_dof[idx], di = depth[i], ai = madr[i];
        for (int k = i + 1; k <= i + ndesc[i]; k++) {
          int ak = madr[k], dk = depth[k] - di;
          float t = W[off + ak + dk] * W[off + ak];  // L(k,i) * D_k
          for (int c = 0; c <= di; c++) W[off + ai + c] -= t * W[off + ak + dk + c];
        }
        float D = W[off + ai];
        if (D < MINVALF) D = MINVALF;
        float inv = 1.f / D;
        W[doff + i] = inv;
        for (int c = 1; c <= di; c++) W[off + ai + c] *= inv;
      }
      GSYNC();
    }
  }

  // x <- (L'DL)^-1 x, x at W[xoff..]
  __device__ __forceinline__ void solve(int off, int doff, int xoff) {
    const int *madr = MI_(DOF_MADR), *ndesc = AUXI(dof_ndesc), *depth = AUXI(dof_depth), *dpar = MI_(DOF_PARENTID);
    const int *dl_adr = MI_(DOF_LEVEL_ADR), *dl_dof = MI_(DOF_LEVEL_DOF);
    for (int lv = a.d.ndoflevel - 2; lv >= 0; lv--) {
      for (int idx = dl_adr[lv] + g; idx < dl_adr[lv + 1]; idx += G) {
        int j = dl_dof[idx], dj = depth[j];
        float s = W[xoff + j];
        for (int k = j + 1; k <= j + ndesc[j]; k++) s -= W[off + madr[k] + depth[k] - dj] * W[xoff + k];
        W[xoff + j] = s;
      }
      GSYNC();
    }
    for (int i = g; i < a.d.nv; i += G) W[xoff + i] *= W[doff + i];
    GSYNC();
    for (int lv = 1; lv < a.d.ndoflevel; lv++) {
      for (int idx = dl_adr[lv] + g; idx < dl_adr[lv + 1]; idx += G) {
        int i = dl_dof[idx];
        float s = W[xoff + i];
        int j = dpar[i], c = 1;
        while (j >= 0) { s -= W[off + madr[i] + c] * W[xoff + j]; j = dpar[j]; c++; }
        W[xoff + i] = s;
      }
      GSYNC();
    }
  }

  // y = M x
  __device__ __forceinline__ void mul_m(int yoff, int xoff) {
    const int *madr = MI_(DOF_MADR), *ndesc = AUXI(dof_ndesc), *depth = AUXI(dof_depth), *dpar = MI_(DOF_PARENTID);
    const Layout& L = a.L;
    for (int i = g; i < a.d.nv; i += G) {
      float s = 0.f;
      int j = i, c = 0;
      while (j >= 0) { s += W[L.qM + madr[i] + c] * W[xoff + j]; j = dpar[j]; c++; }
      int di = depth[i];
      for (int k = i + 1; k <= i + ndesc[i]; k++) s += W[L.qM + madr[k] + depth[k] - di] * W[xoff + k];
      W[yoff + i] = s;
    }
    GSYNC();
  }

  // ------------------------------------------- A5/A6 passive + actuation -> qfrc_smooth
  __device__ __forceinline__ void passive_actuation() {
    const Layout& L = a.L;
    // tendon-level forces: spring/damper + actuators on tendon transmissions
    for (int t = g; t < a.d.ntendon; t += G) {
      float k = MF_(TENDON_STIFFNESS)[t], bd = MF_(TENDON_DAMPING)[t], f = 0.f;
      if (k != 0.f || bd != 0.f) {
        float len = W[L.tenlen + t], lo = MF_(TENDON_LENGTHSPRING)[2 * t], hi = MF_(TENDON_LENGTHSPRING)[2 * t + 1];
        if (len > hi) f = k * (hi - len);
        else if (len < lo) f = k * (lo - len);
        f -= bd * W[L.tenvel + t];
      }
      W[L.tenfrc + t] = f;
    }
    for (int i = g; i < a.d.nv; i += G) W[L.tmp + i] = 0.f;  // joint-transmission actuator forces
    GSYNC();
    for (int u = g; u < a.d.nu; u += G) {
      float ctrl = W[L.ctrl + u];
      if (MI_(ACT_CTRLLIMITED)[u]) ctrl = clampf(ctrl, MF_(ACT_CTRLRANGE)[2 * u], MF_(ACT_CTRLRANGE)[2 * u + 1]);
      int aa = MI_(ACT_ACTADR)[u], id = MI_(ACT_TRNID)[u];
      float gear = MF_(ACT_GEAR)[u], len, vel, input = ctrl;
      bool ten = MI_(ACT_TRNTYPE)[u] == MM_TRN_TENDON;
      if (ten) { len = gear * W[L.tenlen + id]; vel = gear * W[L.tenvel + id]; }
      else { len = gear * W[L.qpos + MI_(JNT_QPOSADR)[id]]; vel = gear * W[L.qvel + MI_(JNT_DOFADR)[id]]; }
      if (MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) {
        float act = W[L.act + aa];
        W[L.actdot + aa] = muscle_dynamics(ctrl, act, MF_(ACT_DYNPRM) + 3 * u);
        input = act;
      }
      float lr0 = MF_(ACT_LENGTHRANGE)[2 * u], lr1 = MF_(ACT_LENGTHRANGE)[2 * u + 1], acc0 = MF_(ACT_ACC0)[u];
      float gain, bias = 0.f;
      if (MI_(ACT_GAINTYPE)[u] == MM_GAIN_MUSCLE) gain = muscle_gain(len, vel, lr0, lr1, acc0, MF_(ACT_GAINPRM) + 9 * u);
      else gain = MF_(ACT_GAINPRM)[9 * u];
      if (MI_(ACT_BIASTYPE)[u] == MM_BIAS_MUSCLE) bias = muscle_bias(len, lr0, lr1, acc0, MF_(ACT_BIASPRM) + 9 * u);
      float f = gain * input + bias;
      if (MI_(ACT_FORCELIMITED)[u]) f = clampf(f, MF_(ACT_FORCERANGE)[2 * u], MF_(ACT_FORCERANGE)[2 * u + 1]);
      W[L.actfrc + u] = f; W[L.actlen + u] = len; W[L.actvel + u] = vel;
      if (ten) atomicAdd(&W[L.tenfrc + id], gear * f);
      else atomicAdd(&W[L.tmp + MI_(JNT_DOFADR)[id]], gear * f);
    }
    GSYNC();
    // qfrc_smooth = passive - bias + actuator  (tendon part gathered through the transposed J)
    const int *ja = AUXI(dofj_adr), *je = AUXI(dofj_entry), *jt = AUXI(dofj_tendon);
    for (int i = g; i < a.d.nv; i += G) {
      float s = -MF_(DOF_DAMPING)[i] * W[L.qvel + i] - W[L.bias + i] + W[L.tmp + i];
      int j = MI_(DOF_JNTID)[i];
      float ks = MF_(JNT_STIFFNESS)[j];
      int type = MI_(JNT_TYPE)[j];
      if (ks != 0.f && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) {
        int qa = MI_(JNT_QPOSADR)[j];
        s -= ks * (W[L.qpos + qa] - MF_(QPOS_SPRING)[qa]);
      }
      for (int e = ja[i]; e < ja[i + 1]; e++) s += W[L.tenj + je[e]] * W[L.tenfrc + jt[e]];
      W[L.smooth + i] = s;
      W[L.qaccsm + i] = s;
    }
    GSYNC();
  }

  // ------------------------------------------------------------- A7 constraint rows
  // row kinds: bits 0-7 type, bit 8 = upper side.  Only rows with a single non-zero of
  // the Jacobian (joint limits) are supported by the sparse Newton path of this engine.
  __device__ __forceinline__ void make_constraint() {
    const Layout& L = a.L;
    int n = 0;
    int nitem = 2 * a.d.njnt;
    for (int base = 0; base < nitem; base += G) {
      int it = base + g;
      bool on = false;
      float dist = 0.f, margin = 0.f;
      int j = it >> 1, side = it & 1;
      if (it < nitem) {
        int type = MI_(JNT_TYPE)[j];
        if (MI_(JNT_LIMITED)[j] && (type == MM_JNT_HINGE || type == MM_JNT_SLIDE)) {
          float q = W[L.qpos + MI_(JNT_QPOSADR)[j]];
          margin = MF_(JNT_MARGIN)[j];
          dist = side == 0 ? q - MF_(JNT_RANGE)[2 * j] : MF_(JNT_RANGE)[2 * j + 1] - q;
          on = dist < margin;
        }
      }
      unsigned long long m = __ballot(on);
      int lane = threadIdx.x & 63;
      int gbase = lane - g;
      unsigned long long gm = (m >> gbase) & (G == 64 ? ~0ull : ((1ull << G) - 1ull));
      int before = __popcll(gm & ((1ull << g) - 1ull));
      int cnt = __popcll(gm);
      if (on) {
        int r = n + before;
        if (r < a.d.njmax) {
          // impedance / reference (solref, solimp) -- MuJoCo constraint model
          const float* si = MF_(JNT_SOLIMP) + 5 * j; const float* sr = MF_(JNT_SOLREF) + 2 * j;
          int dof = MI_(JNT_DOFADR)[j];
          float x = dist - margin;
          float D, aref;
          float Jv = (side == 0 ? 1.f : -1.f) * W[L.qvel + dof];
          impedance(si, sr, x, MF_(DOF_INVWEIGHT0)[dof], Jv, D, aref);
          W[L.efc_kind + r] = __int_as_float(MM_CON_LIMIT_JOINT | (side << 8));
          W[L.efc_id + r] = __int_as_float(dof);
          W[L.efc_pos + r] = x;
          W[L.efc_D + r] = D;
          W[L.efc_aref + r] = aref;
        }
      }
      n += cnt;
    }
    if (n > a.d.njmax) { n = a.d.njmax; status |= 2; }
    nefc = n;
    GSYNC();
  }

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
        else if (xa <= mid) y = powf(xa, power) / powf(mid, power - 1.f);
        else y = 1.f - powf(1.f - xa, power) / powf(1.f - mid, power - 1.f);
        imp = dmin + y * (dmax - dmin);
      }
    }
    float R = fmaxf(MINVALF, (1.f - imp) * diagApprox / imp);
    float K, B;
    if (sr[0] > 0.f) {
      float tc = fmaxf(sr[0], 2.f * a.d.timestep), dr = sr[1];
      K = 1.f / fmaxf(MINVALF, dmax * dmax * tc * tc * dr * dr);
      B = 2.f / fmaxf(MINVALF, dmax * tc);
    } else { K = -sr[0] / fmaxf(MINVALF, dmax * dmax); B = -sr[1] / fmaxf(MINVALF, dmax); }
    D = 1.f / R;
    aref = -B * vel - K * imp * x;
  }

  // J_r . x for the sparse row kinds
  __device__ __forceinline__ float row_dot(int r, int xoff) const {
    int kind = __float_as_int(W[a.L.efc_kind + r]), dof = __float_as_int(W[a.L.efc_id + r]);
    float sgn = (kind >> 8) & 1 ? -1.f : 1.f;
    return sgn * W[xoff + dof];
  }

  // constraint cost + jar for the vector at xoff (collective); returns total cost
  __device__ __forceinline__ float eval_cost(int xoff, bool write_jar) {
    const Layout& L = a.L;
    mul_m(L.Ma, xoff);
    float gs = 0.f;
    for (int i = g; i < a.d.nv; i += G) gs += (W[xoff + i] - W[L.qaccsm + i]) * (W[L.Ma + i] - W[L.smooth + i]);
    float c = 0.5f * gs;
    for (int r = g; r < nefc; r += G) {
      float jar = row_dot(r, xoff) - W[L.efc_aref + r];
      if (write_jar) W[L.efc_jar + r] = jar;
      if (jar < 0.f) c += 0.5f * W[L.efc_D + r] * jar * jar;
    }
    c = gsum<G>(c);
    GSYNC();
    return c;
  }

  struct LsP { float cost, d1, d2; };
  __device__ __forceinline__ LsP ls_eval(float alpha, float q0, float q1, float q2) const {
    const Layout& L = a.L;
    float c = 0.f, d1 = 0.f, d2 = 0.f;
    for (int r = g; r < nefc; r += G) {
      float jv = W[L.efc_jv + r];
      float x = W[L.efc_jar + r] + alpha * jv;
      if (x < 0.f) {
        float D = W[L.efc_D + r];
        c += 0.5f * D * x * x; d1 += D * x * jv; d2 += D * jv * jv;
      }
    }
    LsP p;
    p.cost = gsum<G>(c) + q0 + alpha * (q1 + alpha * q2);
    p.d1 = gsum<G>(d1) + q1 + 2.f * alpha * q2;
    p.d2 = gsum<G>(d2) + 2.f * q2;
    return p;
  }

  __device__ __forceinline__ void update_forces() {
    const Layout& L = a.L;
    for (int i = g; i < a.d.nv; i += G) W[L.qfrccon + i] = 0.f;
    GSYNC();
    for (int r = g; r < nefc; r += G) {
      float jar = W[L.efc_jar + r];
      float f = jar < 0.f ? -W[L.efc_D + r] * jar : 0.f;
      W[L.efc_frc + r] = f;
      if (f != 0.f) {
        int kind = __float_as_int(W[L.efc_kind + r]), dof = __float_as_int(W[L.efc_id + r]);
        atomicAdd(&W[L.qfrccon + dof], ((kind >> 8) & 1 ? -1.f : 1.f) * f);
      }
    }
    GSYNC();
  }

  // Newton solver (primal) with exact line search; mirrors oracle/mmo_engine.c mmo_solve
  __device__ __forceinline__ void solve_constraints() {
    const Layout& L = a.L;
    const int nv = a.d.nv;
    niter = 0;
    if (nefc == 0) {
      for (int i = g; i < nv; i += G) { W[L.qacc + i] = W[L.qaccsm + i]; W[L.qfrccon + i] = 0.f; }
      GSYNC();
      return;
    }
    float scale = 1.f / (a.d.meaninertia * (float)(nv > 1 ? nv : 1));
    // warm start: qacc_warmstart is kept only if it beats the unconstrained solution
    float cost = 0.f, cost_ws = 0.f;
    for (int pass = 0; pass < 3; pass++) {
      int xoff = pass == 0 ? L.warm : (pass == 1 ? L.qaccsm : L.qacc);
      float c = eval_cost(xoff, pass == 2);
      if (pass == 0) cost_ws = c;
      else if (pass == 1) {
        int src = cost_ws < c ? L.warm : L.qaccsm;
        for (int i = g; i < nv; i += G) W[L.qacc + i] = W[src + i];
        GSYNC();
      } else cost = c;
    }
    const int* madr = MI_(DOF_MADR);
    for (int iter = 0; iter < a.d.iterations; iter++) {
      update_forces();
      float gn = 0.f;
      for (int i = g; i < nv; i += G) {
        float gr = W[L.Ma + i] - W[L.smooth + i] - W[L.qfrccon + i];
        W[L.grad + i] = gr; W[L.search + i] = gr;
        gn += gr * gr;
      }
      gn = sqrtf(gsum<G>(gn));
      GSYNC();
      if (scale * gn < a.d.tolerance) break;
      // H = M + sum_active D e_dof e_dof'   (tree sparsity preserved)
      for (int k = g; k < a.d.nM; k += G) W[L.qH + k] = W[L.qM + k];
      GSYNC();
      for (int r = g; r < nefc; r += G)
        if (W[L.efc_jar + r] < 0.f) atomicAdd(&W[L.qH + madr[__float_as_int(W[L.efc_id + r])]], W[L.efc_D + r]);
      GSYNC();
      factor(L.qH, L.hdinv);
      solve(L.qH, L.hdinv, L.search);
      float sn = 0.f;
      for (int i = g; i < nv; i += G) { float s = -W[L.search + i]; W[L.search + i] = s; sn += s * s; }
      sn = sqrtf(gsum<G>(sn));
      GSYNC();
      if (sn < MINVALF) break;
      mul_m(L.Mv, L.search);
      for (int r = g; r < nefc; r += G) W[L.efc_jv + r] = row_dot(r, L.search);
      float q0 = 0.f, q1 = 0.f, q2 = 0.f;
      for (int i = g; i < nv; i += G) {
        float dm = W[L.Ma + i] - W[L.smooth + i], s = W[L.search + i];
        q0 += 0.5f * (W[L.qacc + i] - W[L.qaccsm + i]) * dm;
        q1 += s * dm;
        q2 += 0.5f * s * W[L.Mv + i];
      }
      q0 = gsum<G>(q0); q1 = gsum<G>(q1); q2 = gsum<G>(q2);
      GSYNC();
      float gtol = a.d.tolerance * a.d.ls_tolerance * sn / scale;
      float alpha = 0.f, lo = 0.f, hi = -1.f;
      LsP p = ls_eval(0.f, q0, q1, q2);
      float best_alpha = 0.f, best_cost = p.cost;
      for (int it = 0; it < a.d.ls_iterations; it++) {
        if (fabsf(p.d1) < gtol) break;
        if (p.d1 < 0.f) lo = alpha; else hi = alpha;
        float next = alpha - p.d1 / fmaxf(p.d2, MINVALF);
        if (hi >= 0.f && (next <= lo || next >= hi)) next = 0.5f * (lo + hi);
        else if (hi < 0.f && next <= lo) next = 2.f * lo + 1e-10f;
        if (next == alpha) break;
        alpha = next;
        p = ls_eval(alpha, q0, q1, q2);
        if (p.cost < best_cost) { best_cost = p.cost; best_alpha = alpha; }
      }
      alpha = best_alpha;
      if (alpha == 0.f) break;
      for (int i = g; i < nv; i += G) { W[L.qacc + i] += alpha * W[L.search + i]; W[L.Ma + i] += alpha * W[L.Mv + i]; }
      for (int r = g; r < nefc; r += G) W[L.efc_jar + r] += alpha * W[L.efc_jv + r];
      GSYNC();
      float old = cost;
      cost = best_cost;
      niter = iter + 1;
      if (scale * (old - cost) < a.d.tolerance) { update_forces(); break; }
      if (iter == a.d.iterations - 1) { update_forces(); status |= 4; }
    }
  }

  // ------------------------------------------------------------------ pipeline
  __device__ __forceinline__ void forward() {
    kinematics();
    com_pos();
    tendon();
    make_constraint();
    velocity_bias();
    crb();
    for (int k = g; k < a.d.nM; k += G) W[a.L.qLD + k] = W[a.L.qM + k];
    GSYNC();
    factor(a.L.qLD, a.L.dinv);
    passive_actuation();
    solve(a.L.qLD, a.L.dinv, a.L.qaccsm);
    solve_constraints();
  }

  __device__ __forceinline__ bool bad_state(bool check_acc) {
    const Layout& L = a.L;
    int bad = 0;
    for (int i = g; i < a.d.nq; i += G) bad |= !(fabsf(W[L.qpos + i]) < 1e10f);
    for (int i = g; i < a.d.nv; i += G) {
      bad |= !(fabsf(W[L.qvel + i]) < 1e10f);
      if (check_acc) bad |= !(fabsf(W[L.qacc + i]) < 1e10f);
    }
    return gor<G>(bad) != 0;
  }
  __device__ __forceinline__ void reset_data() {
    const Layout& L = a.L;
    for (int i = g; i < a.d.nq; i += G) W[L.qpos + i] = MF_(QPOS0)[i];
    for (int i = g; i < a.d.nv; i += G) { W[L.qvel + i] = 0.f; W[L.warm + i] = 0.f; }
    for (int i = g; i < a.d.na; i += G) W[L.act + i] = 0.f;
    GSYNC();
  }

  // A9 semi-implicit Euler with implicit joint damping
  __device__ __forceinline__ void euler(float& time) {
    const Layout& L = a.L;
    const float h = a.d.timestep;
    const int* madr = MI_(DOF_MADR);
    int src = L.qacc;
    for (int i = g; i < a.d.nv; i += G) W[L.warm + i] = W[L.qacc + i];
    if (a.d.any_damping && a.d.eulerdamp) {
      for (int k = g; k < a.d.nM; k += G) W[L.qH + k] = W[L.qM + k];
      GSYNC();
      for (int i = g; i < a.d.nv; i += G) {
        W[L.qH + madr[i]] += h * MF_(DOF_DAMPING)[i];
        W[L.tmp + i] = W[L.smooth + i] + W[L.qfrccon + i];
      }
      GSYNC();
      factor(L.qH, L.hdinv);
      solve(L.qH, L.hdinv, L.tmp);
      src = L.tmp;
    }
    for (int u = g; u < a.d.nu; u += G) {
      int aa = MI_(ACT_ACTADR)[u];
      if (aa < 0) continue;
      float x = W[L.act + aa] + h * W[L.actdot + aa];
      if (MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) x = clampf(x, 0.f, 1.f);
      W[L.act + aa] = x;
    }
    for (int i = g; i < a.d.nv; i += G) W[L.qvel + i] += h * W[src + i];
    GSYNC();
    for (int j = g; j < a.d.njnt; j += G) {
      int type = MI_(JNT_TYPE)[j], qa = MI_(JNT_QPOSADR)[j], da = MI_(JNT_DOFADR)[j];
      if (type == MM_JNT_HINGE || type == MM_JNT_SLIDE) { W[L.qpos + qa] += h * W[L.qvel + da]; continue; }
      if (type == MM_JNT_FREE) {
        for (int k = 0; k < 3; k++) W[L.qpos + qa + k] += h * W[L.qvel + da + k];
        qa += 3; da += 3;
      }
      V3 w = ld3(W + L.qvel + da);
      float nw = sqrtf(dot(w, w)), ang = h * nw;
      if (ang > MINVALF) {
        float sn, cs;
        sincosf(0.5f * ang, &sn, &cs);
        float is = sn / nw;
        Q4 dq = {cs, w.x * is, w.y * is, w.z * is};
        Q4 qn = qnorm(qmul(ldq(W + L.qpos + qa), dq));
        W[L.qpos + qa] = qn.w; W[L.qpos + qa + 1] = qn.x; W[L.qpos + qa + 2] = qn.y; W[L.qpos + qa + 3] = qn.z;
      }
    }
    time += h;
    GSYNC();
  }

  // `nsub` mj_step substeps (forward + Euler, MuJoCo bad-state auto-reset semantics) followed by an
  // optional mj_forward on the final state.  One call site of forward() keeps the code size bounded.
  __device__ __forceinline__ void run(int nsub, bool final_forward, float& time) {
    int total = nsub + (final_forward ? 1 : 0);
    int s = 0;
    bool redo = false;
    while (s < total) {
      const bool stepping = s < nsub;
      if (stepping && !redo && bad_state(false)) { reset_data(); time = 0.f; status |= 1; }
      forward();
      if (stepping) {
        if (!redo && bad_state(true)) { reset_data(); time = 0.f; status |= 1; redo = true; continue; }
        euler(time);
        redo = false;
      }
      s++;
    }
  }
};

// =========================================================================== kernels
template <int G>
__global__ void __launch_bounds__(256) k_engine(KArgs a) {
  extern __shared__ float lds[];
  constexpr int EPW = 64 / G;  // envs per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  const int g = lane % G;
  int e = (blockIdx.x * wpb + wave) * EPW + lane / G;
  const int nenv = a.s.nenv;
  if ((blockIdx.x * wpb + wave) * EPW >= nenv) return;  // whole wave idle
  const bool dup = e >= nenv;
  if (dup) e = nenv - 1;  // surplus groups recompute the last env (identical stores)
  float* W = lds + (size_t)(wave * EPW + lane / G) * a.L.total;
  const Layout& L = a.L;
  const Dims& d = a.d;
  Engine<G> E(a, W, g);

  // ---- load state (HBM -> LDS)
  for (int i = g; i < d.nq; i += G) W[L.qpos + i] = a.s.qpos[(size_t)e * d.nq + i];
  for (int i = g; i < d.nv; i += G) {
    W[L.qvel + i] = a.s.qvel[(size_t)e * d.nv + i];
    W[L.warm + i] = a.s.qacc_warmstart[(size_t)e * d.nv + i];
  }
  for (int i = g; i < d.na; i += G) W[L.act + i] = a.s.act[(size_t)e * d.na + i];
  float time = a.s.time[e];
  E.status = a.s.status ? a.s.status[e] : 0;
  const mm_task& t = a.t;
  // ---- action -> ctrl (BaseV0.step: base_v0.py:82-108)
  for (int u = g; u < d.nu; u += G) {
    float c = a.ctrl ? a.ctrl[(size_t)e * d.nu + u] : 0.f;
    if (a.mode == 2 && t.normalize_act && MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) c = 1.f / (1.f + expf(-5.f * (c - 0.5f)));
    if (a.mode == 2 && t.fatigue && MI_(ACT_DYNTYPE)[u] == MM_DYN_MUSCLE) {
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
  if (a.mode == 2 && t.reaf_src >= 0 && t.reaf_dst >= 0 && g == 0) {  // base_v0.py:104-108
    W[L.ctrl + t.reaf_dst] = W[L.ctrl + t.reaf_src];
    W[L.ctrl + t.reaf_src] = 0.f;
  }
  GSYNC();
  if (a.mode == 2 && t.ctrl_out && !dup)
    for (int u = g; u < d.nu; u += G) t.ctrl_out[(size_t)e * d.nu + u] = W[L.ctrl + u];

  int nsub = a.mode == 1 ? 0 : t.nsubsteps;
  bool fwd = a.mode == 1 || (a.mode == 2 && t.do_forward);
  E.run(nsub, fwd, time);

  // ---- store state (surplus groups never write)
  if (dup) return;
  for (int i = g; i < d.nq; i += G) a.s.qpos[(size_t)e * d.nq + i] = W[L.qpos + i];
  for (int i = g; i < d.nv; i += G) {
    a.s.qvel[(size_t)e * d.nv + i] = W[L.qvel + i];
    a.s.qacc_warmstart[(size_t)e * d.nv + i] = W[L.warm + i];
  }
  for (int i = g; i < d.na; i += G) a.s.act[(size_t)e * d.na + i] = W[L.act + i];
  if (g == 0) { a.s.time[e] = time; if (a.s.status) a.s.status[e] = E.status; }

  // ---- derived outputs of the final forward
  if (fwd && a.has_derived) {
    const mm_derived& o = a.o;
    if (o.xpos) for (int i = g; i < 3 * d.nbody; i += G) o.xpos[(size_t)e * 3 * d.nbody + i] = W[L.xpos + i];
    if (o.xquat) for (int i = g; i < 4 * d.nbody; i += G) o.xquat[(size_t)e * 4 * d.nbody + i] = W[L.xquat + i];
    if (o.xipos) for (int i = g; i < 3 * d.nbody; i += G) o.xipos[(size_t)e * 3 * d.nbody + i] = W[L.xipos + i];
    if (o.cvel) for (int i = g; i < 6 * d.nbody; i += G) o.cvel[(size_t)e * 6 * d.nbody + i] = W[L.cvel + i];
    if (o.subtree_com) for (int i = g; i < 3 * d.nbody; i += G) o.subtree_com[(size_t)e * 3 * d.nbody + i] = W[L.com + i];
    if (o.site_xpos)
      for (int s = g; s < d.nsite; s += G) st3(o.site_xpos + ((size_t)e * d.nsite + s) * 3, E.site_pos(s));
    if (o.geom_xpos)
      for (int s = g; s < d.ngeom; s += G) st3(o.geom_xpos + ((size_t)e * d.ngeom + s) * 3, E.geom_pos(s));
    if (o.actuator_length) for (int i = g; i < d.nu; i += G) o.actuator_length[(size_t)e * d.nu + i] = W[L.actlen + i];
    if (o.actuator_velocity) for (int i = g; i < d.nu; i += G) o.actuator_velocity[(size_t)e * d.nu + i] = W[L.actvel + i];
    if (o.actuator_force) for (int i = g; i < d.nu; i += G) o.actuator_force[(size_t)e * d.nu + i] = W[L.actfrc + i];
    if (o.qacc) for (int i = g; i < d.nv; i += G) o.qacc[(size_t)e * d.nv + i] = W[L.qacc + i];
    if (o.ten_length) for (int i = g; i < d.ntendon; i += G) o.ten_length[(size_t)e * d.ntendon + i] = W[L.tenlen + i];
    if (g == 0 && o.nefc) o.nefc[e] = E.nefc;
    if (g == 0 && o.solver_niter) o.solver_niter[e] = E.niter;
  }
  if (a.dbg) for (int i = g; i < L.total; i += G) a.dbg[(size_t)e * L.total + i] = W[i];

  // ---- task stage: obs_dict / reward_dict (pose_v0.py:100-140), TimeLimit counter
  if (a.mode == 2) {
    int sc = 0;
    if (t.step_count) { sc = t.step_count[e] + 1; }
    if (t.task == MM_TASK_POSE) {
      const float dt = t.obs_dt;
      const int o_err = t.obs_layout == 1 ? d.nq + d.nv + d.na : d.nq + d.nv;
      const int o_act = t.obs_layout == 1 ? d.nq + d.nv : 2 * d.nq + d.nv;
      float err2 = 0.f, act2 = 0.f;
      float* ob = t.obs ? t.obs + (size_t)e * t.obs_dim : nullptr;
      for (int i = g; i < d.nq; i += G) {
        float q = W[L.qpos + i];
        float pe = t.target_jnt_value[(size_t)e * d.nq + i] - q;
        err2 += pe * pe;
        if (ob) { ob[i] = q; ob[o_err + i] = pe; }
      }
      for (int i = g; i < d.nv; i += G) if (ob) ob[d.nq + i] = W[L.qvel + i] * dt;
      for (int i = g; i < d.na; i += G) {
        float x = W[L.act + i];
        act2 += x * x;
        if (ob) ob[o_act + i] = x;
      }
      err2 = gsum<G>(err2); act2 = gsum<G>(act2);
      if (g == 0) {
        float pose_dist = sqrtf(err2), act_mag = sqrtf(act2);
        if (d.na != 0 && t.act_reg_mean) act_mag = act_mag / (float)d.na;
        float r_pose = -pose_dist;
        float r_bonus = (pose_dist < t.pose_thd ? 1.f : 0.f) + (pose_dist < 1.5f * t.pose_thd ? 1.f : 0.f);
        float r_pen = pose_dist > t.far_th ? -1.f : 0.f;
        float r_act = -act_mag;
        bool done = pose_dist > t.far_th;
        if (t.rwd) {
          float* r = t.rwd + (size_t)e * MM_RWD_COUNT;
          r[MM_RWD_POSE] = r_pose; r[MM_RWD_BONUS] = r_bonus; r[MM_RWD_PENALTY] = r_pen; r[MM_RWD_ACT_REG] = r_act;
          r[MM_RWD_SPARSE] = -pose_dist; r[MM_RWD_SOLVED] = pose_dist < t.pose_thd ? 1.f : 0.f;
          r[MM_RWD_DONE] = done ? 1.f : 0.f;
          r[MM_RWD_DENSE] = t.w_pose * r_pose + t.w_bonus * r_bonus + t.w_act_reg * r_act + t.w_penalty * r_pen;
        }
        if (t.done) t.done[e] = done ? 1 : 0;
      }
    }
    if (g == 0) {
      if (t.step_count) t.step_count[e] = sc;
      if (t.truncated) t.truncated[e] = (t.max_episode_steps > 0 && sc >= t.max_episode_steps) ? 1 : 0;
    }
  }
}

// ---- Philox4x32-10 (counter based; the oracle side reproduces it in numpy) -----------
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

__global__ void k_uniform(float* out, size_t n, uint64_t seed, uint64_t stream_id) {
  size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t i = i4 * 4;
  if (i >= n) return;
  uint32_t c[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int k = 0; k < 4 && i + k < n; k++) out[i + k] = u01(c[k]);
}

struct ResetArgs {
  const uint32_t* blob; int qpos0_off; int nq, nv, na, nenv;
  mm_state s; const uint8_t* mask; const float* qpos_src; const float* qvel_src;
  // pose reset
  const float *qlo, *qhi, *tlo, *thi; float* target; int32_t* episode; int32_t* step_count; uint64_t seed;
  int pose, random_qpos;
  float* obs; int obs_dim, obs_layout;
};

__global__ void k_reset(ResetArgs r) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= r.nenv) return;
  if (r.mask && !r.mask[e]) return;
  const float* qpos0 = reinterpret_cast<const float*>(r.blob + r.qpos0_off);
  int ep = 0;
  if (r.pose && r.episode) { ep = r.episode[e]; r.episode[e] = ep + 1; }
  for (int i = 0; i < r.nq; i++) {
    float q = r.qpos_src ? r.qpos_src[(size_t)e * r.nq + i] : qpos0[i];
    if (r.pose) {
      // counter = (i/2, which, env, episode): lane 0/1 -> qpos draw for coordinate i (even/odd), lane 2/3 -> target
      uint32_t c[4] = {(uint32_t)(i >> 1), 0u, (uint32_t)e, (uint32_t)ep};
      philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      float uq = u01(c[i & 1]), ut = u01(c[2 + (i & 1)]);
      if (r.random_qpos) q = r.qlo[i] + (r.qhi[i] - r.qlo[i]) * uq;
      if (r.target) r.target[(size_t)e * r.nq + i] = r.tlo[i] + (r.thi[i] - r.tlo[i]) * ut;
    }
    r.s.qpos[(size_t)e * r.nq + i] = q;
    if (r.pose && r.obs) {  // first observation of the new episode: qvel = act = 0
      float* ob = r.obs + (size_t)e * r.obs_dim;
      ob[i] = q;
      ob[(r.obs_layout == 1 ? r.nq + r.nv + r.na : r.nq + r.nv) + i] = r.target[(size_t)e * r.nq + i] - q;
    }
  }
  if (r.pose && r.obs) {
    float* ob = r.obs + (size_t)e * r.obs_dim;
    for (int i = 0; i < r.nv; i++) ob[r.nq + i] = 0.f;
    for (int i = 0; i < r.na; i++) ob[(r.obs_layout == 1 ? r.nq + r.nv : 2 * r.nq + r.nv) + i] = 0.f;
  }
  for (int i = 0; i < r.nv; i++) {
    r.s.qvel[(size_t)e * r.nv + i] = r.qvel_src ? r.qvel_src[(size_t)e * r.nv + i] : 0.f;
    r.s.qacc_warmstart[(size_t)e * r.nv + i] = 0.f;
  }
  for (int i = 0; i < r.na; i++) r.s.act[(size_t)e * r.na + i] = 0.f;
  r.s.time[e] = 0.f;
  if (r.s.status) r.s.status[e] = 0;
  if (r.step_count) r.step_count[e] = 0;
}

// =========================================================================== host side
struct mm_model {
  uint32_t* d_blob = nullptr;
  std::vector<uint32_t> h_blob;
  int sec[MM_NSEC];
  Dims d;
  Layout L;
  Aux x;
  int lanes = 16;
  int waves_per_block = 1;
  size_t lds_per_env = 0;
  int device = 0;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                 \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) return fail(MM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
  } while (0)

extern "C" const char* mm_last_error(void) { return g_err.c_str(); }
extern "C" const char* mm_version(void) { return "myosim-hip 0.1 (gfx950)"; }

static void build_layout(mm_model* m) {
  const Dims& d = m->d;
  Layout& L = m->L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n > 0 ? n : 0); return r; };
  L.qpos = take(d.nq); L.qvel = take(d.nv); L.act = take(d.na); L.ctrl = take(d.nu); L.warm = take(d.nv);
  L.xpos = take(3 * d.nbody); L.xquat = take(4 * d.nbody); L.xmat = take(9 * d.nbody); L.xipos = take(3 * d.nbody);
  L.xanchor = take(3 * d.njnt); L.xaxis = take(3 * d.njnt); L.com = take(3 * d.nbody);
  L.cinert = take(10 * d.nbody); L.cdof = take(6 * d.nv); L.cdofdot = take(6 * d.nv);
  L.cvel = take(6 * d.nbody); L.cacc = take(6 * d.nbody);
  L.tenlen = take(d.ntendon); L.tenvel = take(d.ntendon); L.tenj = take(d.ntenJ); L.tenfrc = take(d.ntendon);
  L.actlen = take(d.nu); L.actvel = take(d.nu); L.actfrc = take(d.nu); L.actdot = take(d.na);
  L.qM = take(d.nM); L.qLD = take(d.nM); L.qH = take(d.nM); L.dinv = take(d.nv); L.hdinv = take(d.nv);
  L.bias = take(d.nv); L.passive = take(0); L.smooth = take(d.nv); L.qaccsm = take(d.nv); L.qacc = take(d.nv);
  L.qfrccon = take(d.nv); L.Ma = take(d.nv); L.grad = take(d.nv); L.search = take(d.nv); L.Mv = take(d.nv);
  L.tmp = take(d.nv);
  int nj = d.njmax > 0 ? d.njmax : 1;
  L.efc_kind = take(nj); L.efc_id = take(nj); L.efc_pos = take(nj); L.efc_D = take(nj); L.efc_aref = take(nj);
  L.efc_jar = take(nj); L.efc_jv = take(nj); L.efc_frc = take(nj);
  // odd stride: neighbouring envs of a wave start on different LDS banks
  if ((o & 1) == 0) o++;
  L.total = o;
  m->lds_per_env = (size_t)o * 4;
}

extern "C" int mm_model_create(const uint32_t* blob, int nwords, mm_model** out) {
  if (!blob || !out || nwords < MM_HEADER_WORDS + 2 * MM_NSEC) return fail(MM_EBADBLOB, "blob too short");
  if (blob[0] != MM_MAGIC || blob[1] != MM_VERSION || blob[2] != MM_NSEC || (int)blob[3] != nwords)
    return fail(MM_EBADBLOB, "bad magic/version/section count");
  mm_model* m = new mm_model();
  m->h_blob.assign(blob, blob + nwords);
  int len[MM_NSEC];
  for (int s = 0; s < MM_NSEC; s++) { m->sec[s] = (int)blob[MM_HEADER_WORDS + 2 * s]; len[s] = (int)blob[MM_HEADER_WORDS + 2 * s + 1]; }
  const int32_t* oi = (const int32_t*)(blob + m->sec[MM_SEC_OPT_I]);
  const float* of = (const float*)(blob + m->sec[MM_SEC_OPT_F]);
  Dims& d = m->d;
  d.nq = oi[MM_OI_NQ]; d.nv = oi[MM_OI_NV]; d.nu = oi[MM_OI_NU]; d.na = oi[MM_OI_NA]; d.nbody = oi[MM_OI_NBODY];
  d.njnt = oi[MM_OI_NJNT]; d.ngeom = oi[MM_OI_NGEOM]; d.nsite = oi[MM_OI_NSITE]; d.ntendon = oi[MM_OI_NTENDON];
  d.nwrap = oi[MM_OI_NWRAP]; d.neq = oi[MM_OI_NEQ]; d.npair = oi[MM_OI_NPAIR]; d.nM = oi[MM_OI_NM];
  d.nlevel = oi[MM_OI_NLEVEL]; d.njmax = oi[MM_OI_NJMAX]; d.ntenJ = oi[MM_OI_NTENJ];
  d.ndoflevel = len[MM_SEC_DOF_LEVEL_ADR] > 0 ? len[MM_SEC_DOF_LEVEL_ADR] - 1 : 0;
  d.iterations = oi[MM_OI_ITERATIONS]; d.ls_iterations = oi[MM_OI_LS_ITERATIONS]; d.eulerdamp = oi[MM_OI_EULERDAMP];
  d.timestep = of[MM_OF_TIMESTEP]; d.gx = of[MM_OF_GRAV_X]; d.gy = of[MM_OF_GRAV_Y]; d.gz = of[MM_OF_GRAV_Z];
  d.tolerance = of[MM_OF_TOLERANCE]; d.ls_tolerance = of[MM_OF_LS_TOLERANCE]; d.meaninertia = of[MM_OF_MEANINERTIA];
  if (oi[MM_OI_INTEGRATOR] != 0) { delete m; return fail(MM_EUNSUPPORTED, "only the Euler integrator is implemented"); }
  if (d.neq > 0 || d.npair > 0) { delete m; return fail(MM_EUNSUPPORTED, "equality/contact rows not implemented in this build"); }
  const int32_t* tlim = (const int32_t*)(blob + m->sec[MM_SEC_TENDON_LIMITED]);
  for (int t = 0; t < d.ntendon; t++)
    if (tlim[t]) { delete m; return fail(MM_EUNSUPPORTED, "tendon limits not implemented in this build"); }
  const float* damp = (const float*)(blob + m->sec[MM_SEC_DOF_DAMPING]);
  d.any_damping = 0;
  for (int i = 0; i < d.nv; i++) if (damp[i] > 0.f) d.any_damping = 1;

  // ---- engine-private tables
  const int32_t* dpar = (const int32_t*)(blob + m->sec[MM_SEC_DOF_PARENTID]);
  const int32_t* bpar = (const int32_t*)(blob + m->sec[MM_SEC_BODY_PARENT]);
  std::vector<int32_t> ndesc(d.nv, 0), depth(d.nv, 0);
  for (int i = 0; i < d.nv; i++) depth[i] = dpar[i] >= 0 ? depth[dpar[i]] + 1 : 0;
  for (int i = d.nv - 1; i >= 0; i--) if (dpar[i] >= 0) ndesc[dpar[i]] += ndesc[i] + 1;
  // descendants must be the contiguous range (i, i+ndesc]: true for depth-first dof numbering
  for (int i = 0; i < d.nv; i++)
    for (int k = i + 1; k <= i + ndesc[i]; k++) {
      int j = k; bool ok = false;
      while (j >= 0) { if (j == i) { ok = true; break; } j = dpar[j]; }
      if (!ok) { delete m; return fail(MM_EUNSUPPORTED, "dofs are not numbered depth-first"); }
    }
  const int32_t* tj_adr = (const int32_t*)(blob + m->sec[MM_SEC_TENJ_ADR]);
  const int32_t* tj_dof = (const int32_t*)(blob + m->sec[MM_SEC_TENJ_DOF]);
  std::vector<int32_t> dj_adr(d.nv + 1, 0), dj_entry, dj_tendon;
  for (int i = 0; i < d.nv; i++) {
    dj_adr[i] = (int)dj_entry.size();
    for (int t = 0; t < d.ntendon; t++)
      for (int e = tj_adr[t]; e < tj_adr[t + 1]; e++)
        if (tj_dof[e] == i) { dj_entry.push_back(e); dj_tendon.push_back(t); }
  }
  dj_adr[d.nv] = (int)dj_entry.size();
  std::vector<int32_t> roots;
  for (int b = 1; b < d.nbody; b++) if (bpar[b] == 0) roots.push_back(b);
  std::vector<uint32_t> dev(m->h_blob);
  auto append = [&](const std::vector<int32_t>& v) {
    int off = (int)dev.size();
    for (int32_t x : v) dev.push_back((uint32_t)x);
    if (v.empty()) dev.push_back(0);
    return off;
  };
  m->x.dof_ndesc = append(ndesc); m->x.dof_depth = append(depth);
  m->x.dofj_adr = append(dj_adr); m->x.dofj_entry = append(dj_entry); m->x.dofj_tendon = append(dj_tendon);
  m->x.root_list = append(roots); m->x.nroot = (int)roots.size();

  build_layout(m);
  HIPCHK(hipGetDevice(&m->device));
  HIPCHK(hipMalloc((void**)&m->d_blob, dev.size() * sizeof(uint32_t)));
  HIPCHK(hipMemcpy(m->d_blob, dev.data(), dev.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  // default lanes per env by model size
  int work = d.ntendon > d.nv ? d.ntendon : d.nv;
  m->lanes = work <= 8 ? 8 : (work <= 24 ? 16 : 32);
  *out = m;
  return MM_OK;
}

extern "C" void mm_model_destroy(mm_model* m) {
  if (!m) return;
  if (m->d_blob) (void)hipFree(m->d_blob);
  delete m;
}

extern "C" int mm_model_set_lanes(mm_model* m, int lanes) {
  if (!m) return MM_EARG;
  if (lanes == 0) return MM_OK;
  if (lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) return fail(MM_EARG, "lanes must be 4..64 pow2");
  m->lanes = lanes;
  return MM_OK;
}

extern "C" int mm_model_info(const mm_model* m, int which) {
  if (!m) return MM_EARG;
  switch (which) {
    case MM_INFO_NQ: return m->d.nq; case MM_INFO_NV: return m->d.nv; case MM_INFO_NU: return m->d.nu;
    case MM_INFO_NA: return m->d.na; case MM_INFO_NBODY: return m->d.nbody; case MM_INFO_NSITE: return m->d.nsite;
    case MM_INFO_NTENDON: return m->d.ntendon; case MM_INFO_LANES_PER_ENV: return m->lanes;
    case MM_INFO_LDS_BYTES_PER_ENV: return (int)m->lds_per_env; case MM_INFO_ENVS_PER_BLOCK: return (64 / m->lanes) * m->waves_per_block;
    case MM_INFO_NGEOM: return m->d.ngeom; case MM_INFO_WAVES_PER_BLOCK: return m->waves_per_block;
  }
  return MM_EARG;
}

// layout query for debugging / tests: returns offset of a named workspace buffer
extern "C" int mm_debug_layout(const mm_model* m, const char* name) {
  const Layout& L = m->L;
#define LQ(n) if (!strcmp(name, #n)) return L.n;
  LQ(qpos) LQ(qvel) LQ(act) LQ(ctrl) LQ(warm) LQ(xpos) LQ(xquat) LQ(xmat) LQ(xipos) LQ(xanchor) LQ(xaxis) LQ(com)
  LQ(cinert) LQ(cdof) LQ(cdofdot) LQ(cvel) LQ(cacc) LQ(tenlen) LQ(tenvel) LQ(tenj) LQ(tenfrc) LQ(actlen) LQ(actvel)
  LQ(actfrc) LQ(actdot) LQ(qM) LQ(qLD) LQ(qH) LQ(dinv) LQ(hdinv) LQ(bias) LQ(smooth) LQ(qaccsm) LQ(qacc) LQ(qfrccon)
  LQ(Ma) LQ(grad) LQ(search) LQ(Mv) LQ(tmp) LQ(efc_kind) LQ(efc_id) LQ(efc_pos) LQ(efc_D) LQ(efc_aref) LQ(efc_jar)
  LQ(efc_jv) LQ(efc_frc) LQ(total)
#undef LQ
  return -1;
}

static int launch(const mm_model* m, KArgs& a, void* stream) {
  int G = m->lanes;
  int epw = 64 / G;
  int wpb = m->waves_per_block;
  int epb = epw * wpb;
  size_t lds = (size_t)epb * m->lds_per_env;
  if (lds > 160 * 1024) return fail(MM_ELDS, "per-block LDS workspace exceeds 160 KiB");
  int nblocks = (a.s.nenv + epb - 1) / epb;
  dim3 grid(nblocks), block(64 * wpb);
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(GG)                                                                                      \
  case GG: {                                                                                            \
    static bool attr_done_##GG = false;                                                                 \
    if (!attr_done_##GG) {                                                                              \
      HIPCHK(hipFuncSetAttribute((const void*)k_engine<GG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      attr_done_##GG = true;                                                                            \
    }                                                                                                   \
    hipLaunchKernelGGL(k_engine<GG>, grid, block, lds, st, a);                                          \
  } break;
  switch (G) { LAUNCH(4) LAUNCH(8) LAUNCH(16) LAUNCH(32) LAUNCH(64) default: return fail(MM_EARG, "bad lanes"); }
#undef LAUNCH
  HIPCHK(hipGetLastError());
  return MM_OK;
}

static void fill_common(const mm_model* m, KArgs& a, const mm_state* s) {
  memset(&a, 0, sizeof(a));
  a.blob = m->d_blob;
  memcpy(a.sec, m->sec, sizeof(a.sec));
  a.d = m->d; a.L = m->L; a.x = m->x; a.s = *s;
}

static float* g_dbg = nullptr;
extern "C" void mm_debug_set_dump(float* dev_ptr) { g_dbg = dev_ptr; }

extern "C" int mm_step(const mm_model* m, const mm_state* s, const float* ctrl, int nsub, void* stream) {
  if (!m || !s || nsub < 0) return fail(MM_EARG, "mm_step: bad argument");
  KArgs a; fill_common(m, a, s);
  a.ctrl = ctrl; a.mode = 0; a.t.nsubsteps = nsub; a.dbg = nullptr;
  return launch(m, a, stream);
}

extern "C" int mm_forward(const mm_model* m, const mm_state* s, const float* ctrl, const mm_derived* out, void* stream) {
  if (!m || !s) return fail(MM_EARG, "mm_forward: bad argument");
  KArgs a; fill_common(m, a, s);
  a.ctrl = ctrl; a.mode = 1;
  if (out) { a.o = *out; a.has_derived = 1; }
  a.dbg = g_dbg;
  return launch(m, a, stream);
}

extern "C" int mm_env_step(const mm_model* m, const mm_state* s, const float* action, const mm_task* t,
                           const mm_derived* out, void* stream) {
  if (!m || !s || !t) return fail(MM_EARG, "mm_env_step: bad argument");
  if (t->task == MM_TASK_POSE && !t->target_jnt_value) return fail(MM_EARG, "pose task needs target_jnt_value");
  if (t->task != MM_TASK_NONE && t->task != MM_TASK_POSE) return fail(MM_EUNSUPPORTED, "task not implemented");
  if (t->fatigue && (!t->fat_MA || !t->fat_MR || !t->fat_MF)) return fail(MM_EARG, "fatigue needs MA/MR/MF");
  KArgs a; fill_common(m, a, s);
  a.ctrl = action; a.mode = 2; a.t = *t;
  if (out) { a.o = *out; a.has_derived = 1; }
  a.dbg = g_dbg;
  return launch(m, a, stream);
}

extern "C" int mm_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qpos_src,
                        const float* qvel_src, void* stream) {
  if (!m || !s) return fail(MM_EARG, "mm_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.qpos_src = qpos_src; r.qvel_src = qvel_src;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_pose_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qlo,
                             const float* qhi, const float* tlo, const float* thi, float* target, int32_t* episode,
                             int32_t* step_count, uint64_t seed, int random_qpos, float* obs, int obs_dim,
                             int obs_layout, void* stream) {
  if (!m || !s || !tlo || !thi || !target) return fail(MM_EARG, "mm_pose_reset: bad argument");
  if (random_qpos && (!qlo || !qhi)) return fail(MM_EARG, "mm_pose_reset: random_qpos needs qlo/qhi");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask;
  r.qlo = qlo; r.qhi = qhi; r.tlo = tlo; r.thi = thi; r.target = target; r.episode = episode;
  r.step_count = step_count; r.seed = seed; r.pose = 1; r.random_qpos = random_qpos;
  r.obs = obs; r.obs_dim = obs_dim; r.obs_layout = obs_layout;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_uniform(float* out, size_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  if (!out) return fail(MM_EARG, "mm_uniform: null output");
  size_t n4 = (n + 3) / 4;
  hipLaunchKernelGGL(k_uniform, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, stream_id);
  HIPCHK(hipGetLastError());
  return MM_OK;
}
