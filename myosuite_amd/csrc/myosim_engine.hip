// myosim_engine.hip -- host side of the C ABI (include/myosim.h): model upload, LDS layout, kernel selection / launch,
// reset and Philox kernels.  The fused physics kernel template is in myosim_engine_kernel.hpp.
#include <array>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include "myosim_engine_kernel.hpp"
#include "myosim_engine_kernel_f64.hpp"
#include "myosim_inst_list.hpp"
MM_KERNEL_LIST(MM_DECLARE)
MM_KERNELS_OBS(MM_DECLARE_OBS)
namespace mm64 {
MM_KERNELS_F64(MM_DECLARE)
}

// Philox4x32-10 / u01: myosim_engine_kernel.hpp
// out[i] = word (first+i)%4 of Philox counter ((first+i)/4, stream_id): one thread per counter
__global__ void k_uniform(float* out, size_t n, uint64_t seed, uint64_t stream_id, size_t first) {
  const size_t i4 = first / 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 * 4 >= first + n) return;
  uint32_t c[4] = {(uint32_t)i4, (uint32_t)(i4 >> 32), (uint32_t)stream_id, (uint32_t)(stream_id >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  for (int k = 0; k < 4; k++) {
    const size_t gi = i4 * 4 + k;
    if (gi >= first && gi < first + n) out[gi - first] = u01(c[k]);
  }
}

struct ResetArgs {
  const uint32_t* blob; int qpos0_off; int nq, nv, na, nenv;
  mm_state s; const uint8_t* mask; const float* qpos_src; const float* qvel_src; const float* qpos_bcast;
  const float *qlo, *qhi, *tlo, *thi; float* target; int32_t* episode; int32_t* step_count; uint64_t seed;
  int pose, random_qpos;
  float* obs; int obs_dim, obs_layout;
  int reach, ntip; const float* tip0;
  int walk, walk_random; const float *ka_qpos, *ka_qvel, *kb_qpos, *kb_qvel;
  int reor, reor_ntab; const float* reor_tab; float *reor_gsize, *reor_axis_half, *reor_des_rot; float reor_tar_length;
  int32_t* reor_gtype;   // non-null: also draw the object type (tables [4][ntab][3])
  int pen; float pen_axis_half, pen_lo0, pen_hi0, pen_lo1, pen_hi1;   // pen-twirl reset: fixed geometry, euler ranges
  int hold; const float* hold_center; float hold_half, hold_slo, hold_shi; float* hold_goal; float* hold_gsize;
  int state_f64;   // MM_PREC_F64_STATE: the four state rows of mm_state are fp64
};
// one element of a state row (fp32, or fp64 behind the same pointer in precision mode MM_PREC_F64_STATE)
__device__ __forceinline__ void st_row(float* p, size_t i, float v, int f64) {
  if (f64) reinterpret_cast<double*>(p)[i] = (double)v; else p[i] = v;
}

__global__ void k_reset(ResetArgs r) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= r.nenv) return;
  if (r.mask && !r.mask[e]) return;
  const uint32_t ge = (uint32_t)(r.s.env_index_base + e);   // global env index: keys every Philox stream below
  const float* qpos0 = reinterpret_cast<const float*>(r.blob + r.qpos0_off);
  int ep = 0;
  if (r.pose && r.episode) { ep = r.episode[e]; r.episode[e] = ep + 1; }
  for (int i = 0; i < r.nq; i++) {
    float q = r.qpos_src ? r.qpos_src[(size_t)e * r.nq + i] : (r.qpos_bcast ? r.qpos_bcast[i] : qpos0[i]);
    if (r.pose) {
      // counter = (i/2, 0, env, episode): words 0/1 -> qpos draw of coordinate i (even/odd), words 2/3 -> target
      uint32_t c[4] = {(uint32_t)(i >> 1), 0u, ge, (uint32_t)ep};
      philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      float uq = u01(c[i & 1]), ut = u01(c[2 + (i & 1)]);
      if (r.random_qpos) q = r.qlo[i] + (r.qhi[i] - r.qlo[i]) * uq;
      if (r.target) r.target[(size_t)e * r.nq + i] = r.tlo[i] + (r.thi[i] - r.tlo[i]) * ut;
    }
    st_row(r.s.qpos, (size_t)e * r.nq + i, q, r.state_f64);
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
  if (r.reach) {
    int ep = r.episode ? r.episode[e] : 0;
    if (r.episode) r.episode[e] = ep + 1;
    const int n3 = 3 * r.ntip;
    float* ob = r.obs ? r.obs + (size_t)e * r.obs_dim : nullptr;
    for (int i = 0; i < n3; i++) {
      uint32_t c[4] = {(uint32_t)(i >> 2), 1u, ge, (uint32_t)ep};
      philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      float tg = r.tlo[i] + (r.thi[i] - r.tlo[i]) * u01(c[i & 3]);
      r.target[(size_t)e * n3 + i] = tg;
      if (ob) { ob[r.nq + r.nv + i] = r.tip0[i]; ob[r.nq + r.nv + n3 + i] = tg - r.tip0[i]; }
    }
    if (ob) {
      for (int i = 0; i < r.nq; i++) ob[i] = qpos0[i];
      for (int i = 0; i < r.nv; i++) ob[r.nq + i] = 0.f;
      for (int i = 0; i < r.na; i++) ob[r.nq + r.nv + 2 * n3 + i] = 0.f;
    }
  }
  if (r.hold) {
    // obj_hold_v0.py:134-145: goal ~ center + U(-half, half)^3 (counter (0,4,env,episode)), object size ~ U(lo, hi)^3
    // (counter (1,4,env,episode)); Fixed task: half = 0, no size table
    int ep = r.episode ? r.episode[e] : 0;
    if (r.episode) r.episode[e] = ep + 1;
    uint32_t c[4] = {0u, 4u, ge, (uint32_t)ep};
    philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    for (int k = 0; k < 3; k++) r.hold_goal[(size_t)e * 3 + k] = r.hold_center[k] + r.hold_half * (2.f * u01(c[k]) - 1.f);
    if (r.hold_gsize) {
      uint32_t c2[4] = {1u, 4u, ge, (uint32_t)ep};
      philox4x32_10(c2, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      for (int k = 0; k < 3; k++) r.hold_gsize[(size_t)e * 3 + k] = r.hold_slo + (r.hold_shi - r.hold_slo) * u01(c2[k]);
    }
  }
  if (r.reor) {
    // reorient_sar_v0.py:386-432 (capsule branch): Philox counter (0, 3, env, episode): word 0 -> size-table row,
    // words 1 / 2 -> desired_orien[0] ~ U(-1,1), desired_orien[1] ~ U(-0.8,1.2)
    int ep = r.episode ? r.episode[e] : 0;
    if (r.episode) r.episode[e] = ep + 1;
    uint32_t c[4] = {0u, 3u, ge, (uint32_t)ep};
    philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    float ah, e0, e1;
    if (r.pen) {   // pen_v0.py:171-184: fixed geometry, desired_orien[0:2] ~ U(lo, hi)
      ah = r.pen_axis_half;
      e0 = r.pen_lo0 + (r.pen_hi0 - r.pen_lo0) * u01(c[1]); e1 = r.pen_lo1 + (r.pen_hi1 - r.pen_lo1) * u01(c[2]);
    } else {
      int idx = (int)(u01(c[0]) * (float)r.reor_ntab);
      if (idx >= r.reor_ntab) idx = r.reor_ntab - 1;
      int ty = 0;   // 0 capsule, 1 ellipsoid, 2 cylinder, 3 box  (geom types 3..6; word 3 of the counter)
      if (r.reor_gtype) { ty = (int)(u01(c[3]) * 4.f); if (ty > 3) ty = 3; r.reor_gtype[e] = MM_GEOM_CAPSULE + ty; }
      const float* sz = r.reor_tab + 3 * (ty * r.reor_ntab + idx);
      for (int k = 0; k < 3; k++) r.reor_gsize[(size_t)e * 3 + k] = sz[k];
      ah = ty == 0 ? 1.3f * sz[1] : (ty == 2 ? sz[1] : sz[2]);   // reorient_sar_v0.py:390-406
      r.reor_axis_half[e] = ah;
      e0 = -1.f + 2.f * u01(c[1]); e1 = -0.8f + 2.f * u01(c[2]);
    }
    // euler2quat([e0, e1, 0]) (utils/quat_math.py:70-86): ai = 0, aj = -e1/2, ak = e0/2
    const float aj = -0.5f * e1, ak = 0.5f * e0;
    const float sj = sinf(aj), cj = cosf(aj), sk = sinf(ak), ck = cosf(ak);
    const float qw = cj * ck, qx = cj * sk, qy = -(sj * ck), qz = -sj * sk;
    // third column of quat2mat(q) times 2*axis_half / tar_length
    const float sc_ = 2.f * ah / r.reor_tar_length;
    r.reor_des_rot[(size_t)e * 3 + 0] = 2.f * (qx * qz + qw * qy) * sc_;
    r.reor_des_rot[(size_t)e * 3 + 1] = 2.f * (qy * qz - qw * qx) * sc_;
    r.reor_des_rot[(size_t)e * 3 + 2] = (1.f - 2.f * (qx * qx + qy * qy)) * sc_;
  }
  if (r.walk) {
    // walk_v0.py:327-365: key pose (random: coin between the two stride keys + N(0, 0.02) on every coordinate
    // except root height and root quaternion); Philox counters (i, 2, env, episode), coin at i = 0xFFFF
    int ep = r.episode ? r.episode[e] : 0;
    if (r.episode) r.episode[e] = ep + 1;
    const float *kq = r.ka_qpos, *kv = r.ka_qvel;
    if (r.walk_random) {
      uint32_t c[4] = {0xFFFFu, 2u, ge, (uint32_t)ep};
      philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      if (!(u01(c[0]) < 0.5f)) { kq = r.kb_qpos; kv = r.kb_qvel; }
    }
    for (int i = 0; i < r.nq; i++) {
      float q = kq[i];
      if (r.walk_random && !(i >= 2 && i < 7)) {
        uint32_t c[4] = {(uint32_t)i, 2u, ge, (uint32_t)ep};
        philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
        float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = u01(c[1]);
        q += 0.02f * sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
      }
      st_row(r.s.qpos, (size_t)e * r.nq + i, q, r.state_f64);
    }
    for (int i = 0; i < r.nv; i++) st_row(r.s.qvel, (size_t)e * r.nv + i, kv[i], r.state_f64);
  }
  for (int i = 0; i < r.nv; i++) {
    if (!r.walk) st_row(r.s.qvel, (size_t)e * r.nv + i, r.qvel_src ? r.qvel_src[(size_t)e * r.nv + i] : 0.f, r.state_f64);
    st_row(r.s.qacc_warmstart, (size_t)e * r.nv + i, 0.f, r.state_f64);
  }
  for (int i = 0; i < r.na; i++) st_row(r.s.act, (size_t)e * r.na + i, 0.f, r.state_f64);
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
  Layout L, Ltw;   // LDS tables of an env in a one-wave / two-wave launch (env_layout)
  DbgLayout D;
  Aux x;
  int lanes = 64;
  int lanes_auto = 1;        // pick the group width per launch from the batch size
  int lanes_user = 0;        // the width was pinned by the caller (mm_model_set_lanes), not chosen as the model's default
  int nvp = 24;
  int waves_per_block = 0;   // 0 = auto
  int lds_model = 1;
  int blob_words = 0;
  int cofs = 0, cofs_tw = 0; // word offsets of the ConstBlocks (one-wave / two-wave launches) behind the blob (device copy only)
  size_t lds_per_env = 0, lds_per_env_tw = 0;   // bytes of LDS tables per env (one-wave / two-wave launches)
  int device = 0;
  float origin[3] = {0.f, 0.f, 0.f};   // internal world-frame origin (see Dims::ox)
  std::vector<int32_t> desc_all, seg_tab, anc_tab;   // Aux::dof_desc / dof_seg / dof_anc, built with the dims
  int nseg = 0;                             // segments of the dof tree (SP kernels)
  int nwrapitem = 0;                        // tendon path items that wrap a geom (tangent points kept in LDS)
  std::vector<double> ten_len0;             // per tendon: summed length of its path segments between rigidly connected bodies (folded at create)
  std::vector<uint8_t> baked_body;          // bodies whose frame position such a folded segment spans (a per-env body_pos on one is refused)
  int nfolded = 0;                          // path items folded into ten_len0
  int precision = MM_PREC_F32;              // MM_PREC_*: which kernel family steps this model (mm_model_set_option "precision")
};

static int upload_consts(mm_model* m);
static thread_local std::string g_err;
static int g_two_wave = 1;   // MYOSIM_TWO_WAVE=0 switches the helper waves off (A/B, debugging)
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                 \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) return fail(MM_EHIP, std::string(#x) + ": " + hipGetErrorString(e_));   \
  } while (0)

extern "C" const char* mm_last_error(void) { return g_err.c_str(); }
extern "C" const char* mm_version(void) { return "myosim-hip 0.4 (gfx950, lane=item engine, ABI 7)"; }
extern "C" int mm_abi_version(void) { return MM_ABI_VERSION; }
extern "C" int mm_struct_size(int which) {
  switch (which) {
    case MM_STRUCT_STATE: return (int)sizeof(mm_state); case MM_STRUCT_DERIVED: return (int)sizeof(mm_derived);
    case MM_STRUCT_TASK: return (int)sizeof(mm_task); case MM_STRUCT_ROLLOUT: return (int)sizeof(mm_rollout);
  }
  return MM_EARG;
}

static const int kNvpChoices[] = {4, 24, 32, 36, 40};
// integrator -> kernel variant (template argument INTEG)
static int integ_kernel(int integrator) { return integrator == MM_INT_RK4 ? 1 : (integrator == MM_INT_IMPLICITFAST ? 2 : 0); }

// is (lanes_per_env, padded nv, general-rows, integrator) a compiled instantiation?  (myosim_inst_list.hpp)
static bool have_kernel(int G, int nvp, int gen, int rk4 = 0) {
#define X(G_, N_, GN_, RK_) if (G == G_ && nvp == N_ && gen == GN_ && rk4 == RK_) return true;
  MM_KERNEL_LIST(X)
#undef X
  return false;
}

// ... and of the precision-mode family (mm64::k_engine; myosim_inst_list.hpp: MM_KERNELS_F64)
static bool have_kernel_f64(int G, int nvp, int gen, int rk4) {
#define X(G_, N_, GN_, RK_) if (G == G_ && nvp == N_ && gen == GN_ && rk4 == RK_) return true;
  MM_KERNELS_F64(X)
#undef X
  return false;
}
// the check every width decision goes through: a compiled instantiation of the model's kernel family
static bool have_model_kernel(const mm_model* m, int G) {
  const int rk = integ_kernel(m->d.integrator);
  return m->precision != MM_PREC_F32 ? have_kernel_f64(G, m->nvp, m->d.gen, rk) : have_kernel(G, m->nvp, m->d.gen, rk);
}

// LDS tables of one env.  two_wave: the layout of a launch that gives every env a helper wave (Engine::TW): tables that share words
// in a one-wave launch because ONE wave never needs both at a time (joint anchors / axes vs the composite inertias, the tendons'
// tangent points vs the u1 scratch) get their own words, plus a second dense tile and the meeting counters.
static Layout env_layout(const mm_model* m, bool two_wave) {
  const Dims& d = m->d;
  Layout L;
  memset(&L, 0, sizeof(L));
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n > 0 ? n : 0); return r; };
  L.qpos = take(d.nq); L.qvel = take(d.nv); L.act = take(d.na); L.ctrl = take(d.nu); L.actdot = take(d.na);
  L.xpos = take(3 * d.nbody); L.xmat = take(9 * d.nbody);
  L.com = take(3 * m->x.nroot); L.cdof = take(6 * d.nv);
  o = (o + 3) & ~3;
  // 12 words (cvel, cacc) + 1 pointer-jumping word per body | dense tile | SP kernels: published rows [nvp][12], x [nvp], update
  // matrices [nseg][36]
  // row stride of the dense tile(s): Engine::TD (the 32-wide tile of the dense kernels is padded against LDS bank conflicts)
  const bool sp_kernel = MM_SPARSE_LDL && !d.gen && m->nvp >= 8 && d.integrator != MM_INT_IMPLICITFAST;
  const int td = (!sp_kernel && m->nvp == 32) ? 36 : m->nvp;
  const int u1_words = std::max(std::max(14 * d.nbody, m->nvp * td), d.seg_u + 36 * m->nseg);   // (CVS + 1) * nbody: Engine::CVS
  L.u1 = take(u1_words);
  if (two_wave) { L.crb = take(10 * d.nbody); L.xanchor = take(3 * d.njnt); L.xaxis = take(3 * d.njnt); }
  else { L.crb = take(std::max(10 * d.nbody, 6 * d.njnt)); L.xanchor = L.crb; L.xaxis = L.crb + 3 * d.njnt; }   // anchors / axes die before crb
  L.tenlen = take(d.ntendon); L.tenvel = take(d.ntendon); L.tenj = take(d.ntenJ); L.tenfrc = take(d.ntendon);
  L.wrapw = (!two_wave && 7 * m->nwrapitem <= u1_words) ? L.u1 : take(7 * m->nwrapitem);   // u1 is free between FK and the velocity stage
  L.flags = take(two_wave ? 4 : 0);
  L.actlen = take(d.nu); L.actvel = take(d.nu); L.actfrc = take(d.nu);
  L.vec = take(d.nv);
  o = (o + 3) & ~3;
  L.xvec = take(m->nvp);
  if (d.integrator == MM_INT_RK4) { L.rk_qpos0 = take(d.nq); L.rk_act0 = take(d.na); L.rk_adot = take(d.na); }
  if (d.integrator == MM_INT_IMPLICITFAST) { L.tenw = take(d.ntendon); L.dofw = take(d.nv); }
  if (d.gen) { o = (o + 3) & ~3; L.efcJ = take(d.efc_rows * (m->nvp + 4)); L.rowtab = take(3 * m->lanes); }
  if (two_wave) { o = (o + 3) & ~3; L.mtile = take(m->nvp * td + m->nvp); }
  // 16-byte aligned env stride (wide ds_read/ds_write never straddle), skewed by 4 words so that neighbouring
  // envs of a wave do not start on the same LDS bank
  o = (o + 3) & ~3;
#ifndef MM_ENV_SKEW
#define MM_ENV_SKEW 1
#endif
  if (MM_ENV_SKEW && m->nvp <= 4) {
    // tiny models run 4 .. 16 envs per wave (8 lanes per env for the elbow at 4096 envs): a ds_read_b32 is serviced in groups of
    // 32 lanes over 32 banks, i.e. four 8-lane envs at a time -- an env stride of 8 (mod 32) words puts their same-offset
    // accesses on disjoint banks (rocprofv3: 32 % of the elbow kernel's LDS cycles were conflict cycles with the old skew of 4)
    while ((o & 31) != 8) o += 4;
  } else if ((o & 31) == 0) o += 4;
  L.total = o;
  return L;
}
static void build_layout(mm_model* m) {
  m->d.efc_rows = std::min(m->lanes, (m->d.njmax + 3) & ~3);
  m->d.seg_u = 13 * m->nvp;
  const Dims& d = m->d;
  m->L = env_layout(m, false);
  m->Ltw = env_layout(m, true);
  const size_t word = m->precision != MM_PREC_F32 ? 8 : 4;   // the tables hold `real`: precision mode doubles them
  m->lds_per_env = (size_t)m->L.total * word;
  m->lds_per_env_tw = (size_t)m->Ltw.total * word;
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n > 0 ? n : 0); return r; };
  DbgLayout& D = m->D;
  o = 0;
  D.xpos = take(3 * d.nbody); D.xquat = take(4 * d.nbody); D.xipos = take(3 * d.nbody); D.cdof = take(6 * d.nv);
  D.cvel = take(6 * d.nbody); D.tenlen = take(d.ntendon); D.tenvel = take(d.ntendon); D.tenj = take(d.ntenJ);
  D.actfrc = take(d.nu); D.actdot = take(d.na); D.M = take(d.nv * d.nv); D.bias = take(d.nv); D.smooth = take(d.nv);
  D.qaccsm = take(d.nv); D.qacc = take(d.nv); D.qfrccon = take(d.nv);
  D.efc_active = take(64); D.efc_D = take(64); D.efc_aref = take(64); D.scal = take(32 + 64);   // (+ 64: per-iteration Newton trace of a MM_NEWTON_TRACE tools build)
  D.total = o;
}

static int check_lanes(const mm_model* m, int lanes) {
  const Dims& d = m->d;
  if (lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) return 0;
  if (d.nbody > lanes || d.nv > lanes || d.njnt > lanes || m->nvp > lanes) return 0;
  // one constraint row / equality per lane; the explicit pair list is swept in chunks of `lanes` pairs (make_constraint_gen), bounded
  // by MM_MAX_PAIRS (the pair index shares a row-descriptor word with the row kind)
  if (d.gen && (d.njmax > lanes || d.neq > lanes || d.npair > MM_MAX_PAIRS)) return 0;
  return 1;
}

extern "C" int mm_model_create(const uint32_t* blob, int nwords, mm_model** out) {
  { const char* tw = getenv("MYOSIM_TWO_WAVE"); if (tw) g_two_wave = atoi(tw) != 0; }
  if (!blob || !out || nwords < MM_HEADER_WORDS + 2 * MM_NSEC) return fail(MM_EBADBLOB, "blob too short");
  if (blob[0] != MM_MAGIC || blob[1] != MM_VERSION || blob[2] != MM_NSEC || (int)blob[3] != nwords)
    return fail(MM_EBADBLOB, "bad magic/version/section count");
  mm_model* m = new mm_model();
  m->h_blob.assign(blob, blob + nwords);
  int len[MM_NSEC];
  for (int s = 0; s < MM_NSEC; s++) { m->sec[s] = (int)blob[MM_HEADER_WORDS + 2 * s]; len[s] = (int)blob[MM_HEADER_WORDS + 2 * s + 1]; }
  (void)len;
  const int32_t* oi = (const int32_t*)(blob + m->sec[MM_SEC_OPT_I]);
  const float* of = (const float*)(blob + m->sec[MM_SEC_OPT_F]);
  Dims& d = m->d;
  d.nq = oi[MM_OI_NQ]; d.nv = oi[MM_OI_NV]; d.nu = oi[MM_OI_NU]; d.na = oi[MM_OI_NA]; d.nbody = oi[MM_OI_NBODY];
  d.njnt = oi[MM_OI_NJNT]; d.ngeom = oi[MM_OI_NGEOM]; d.nsite = oi[MM_OI_NSITE]; d.ntendon = oi[MM_OI_NTENDON];
  d.nwrap = oi[MM_OI_NWRAP]; d.neq = oi[MM_OI_NEQ]; d.npair = oi[MM_OI_NPAIR]; d.nM = oi[MM_OI_NM];
  d.nlevel = oi[MM_OI_NLEVEL]; d.njmax = oi[MM_OI_NJMAX]; d.nconmax = oi[MM_OI_NCONMAX]; d.ntenJ = oi[MM_OI_NTENJ];
  d.condim4 = 0;     // set below when a pair carries condim 4
  d.iterations = oi[MM_OI_ITERATIONS]; d.ls_iterations = oi[MM_OI_LS_ITERATIONS]; d.eulerdamp = oi[MM_OI_EULERDAMP];
  d.timestep = of[MM_OF_TIMESTEP]; d.gx = of[MM_OF_GRAV_X]; d.gy = of[MM_OF_GRAV_Y]; d.gz = of[MM_OF_GRAV_Z];
  d.tolerance = of[MM_OF_TOLERANCE]; d.ls_tolerance = of[MM_OF_LS_TOLERANCE]; d.meaninertia = of[MM_OF_MEANINERTIA];
  d.integrator = oi[MM_OI_INTEGRATOR];
  if (d.integrator != MM_INT_EULER && d.integrator != MM_INT_RK4 && d.integrator != MM_INT_IMPLICITFAST) {
    delete m; return fail(MM_EUNSUPPORTED, "integrator must be Euler (0), RK4 (1) or implicitfast (3)");
  }
  d.ntlim = 0;
  {
    const int32_t* tlim = (const int32_t*)(blob + m->sec[MM_SEC_TENDON_LIMITED]);
    const float* trng = (const float*)(blob + m->sec[MM_SEC_TENDON_RANGE]);
    const float* tmar = (const float*)(blob + m->sec[MM_SEC_TENDON_MARGIN]);
    for (int t = 0; t < d.ntendon; t++) {
      if (!tlim[t]) continue;
      d.ntlim++;
      if (trng[2 * t + 1] - trng[2 * t] < 2.f * tmar[t]) { delete m; return fail(MM_EUNSUPPORTED, "tendon range narrower than 2*margin"); }
    }
  }
  d.nfric = 0;
  {
    const float* fl = (const float*)(blob + m->sec[MM_SEC_DOF_FRICTIONLOSS]);
    for (int i = 0; i < d.nv; i++) if (fl[i] > 0.f) d.nfric++;
  }
  d.gen = (d.neq > 0 || d.npair > 0 || d.nfric > 0 || d.ntlim > 0) ? 1 : 0;
  {
    // dof-tree depth.  The limit-rows-only kernels keep M tree-sparse with at most 8 entries per row (dof + 7 ancestors);
    // a deeper tree takes the general-row kernels, whose factorisations are dense.
    const int32_t* dpar = (const int32_t*)(blob + m->sec[MM_SEC_DOF_PARENTID]);
    int maxd = 0;
    for (int i = 0; i < d.nv; i++) {
      int dep = 0;
      for (int j = dpar[i]; j >= 0; j = dpar[j]) dep++;
      if (dep > maxd) maxd = dep;
    }
    d.dof_nlevel = maxd + 1;
    // Tables of the tree-sparse factorisation (Engine::sp_factor_solve / sp_mul_m): depth of every dof, its descendants, and the
    // SEGMENTS of the dof tree (maximal unbranched chains; a dof starts a segment when its parent has another child too).
    std::vector<int> dep(d.nv, 0), nchild(d.nv, 0);
    for (int i = 0; i < d.nv; i++) { dep[i] = dpar[i] < 0 ? 0 : dep[dpar[i]] + 1; if (dpar[i] >= 0) nchild[dpar[i]]++; }
    bool fits = d.dof_nlevel <= 8 && d.nv < 255;
    const size_t nvs = (size_t)(d.nv > 0 ? d.nv : 1);
    m->desc_all.assign(nvs * 8, -1);
    m->seg_tab.assign(nvs * 6, -1);
    for (int i = 0; i < d.nv; i++) m->seg_tab[(size_t)i * 6 + 5] = dep[i];
    m->anc_tab.assign(nvs * 2, 0);
    for (int i = 0; i < d.nv && fits; i++)
      for (int k = dpar[i]; k >= 0; k = dpar[k]) m->anc_tab[(size_t)i * 2 + (dep[k] >> 2)] |= (int32_t)((uint32_t)k << (8 * (dep[k] & 3)));
    d.seg_nlevel = 0; d.seg_lvinfo[0] = d.seg_lvinfo[1] = 0; d.seg_lvtb[0] = d.seg_lvtb[1] = 0; d.seg_zero = 0; d.desc_words = 0; m->nseg = 0;
    if (fits) {
      std::vector<int> ndesc(d.nv, 0);
      for (int k = 0; k < d.nv && fits; k++)
        for (int i = dpar[k]; i >= 0; i = dpar[i]) {
          const int c = ndesc[i]++;
          if (c >= 32) { fits = false; break; }
          uint32_t* w = (uint32_t*)&m->desc_all[(size_t)i * 8 + (c >> 2)];
          *w = (*w & ~(255u << (8 * (c & 3)))) | ((uint32_t)k << (8 * (c & 3)));
          d.desc_words = std::max(d.desc_words, (c >> 2) + 1);
        }
    }
    if (fits) {
      struct Seg { int top, bottom, parent, level, nch; int ch[8]; };
      std::vector<Seg> segs;
      std::vector<int> seg_of(d.nv, -1);
      for (int k = 0; k < d.nv && fits; k++) {
        if (dpar[k] >= 0 && nchild[dpar[k]] == 1) { seg_of[k] = seg_of[dpar[k]]; segs[seg_of[k]].bottom = k; continue; }
        Seg sg{}; sg.top = sg.bottom = k; sg.parent = dpar[k] >= 0 ? seg_of[dpar[k]] : -1;
        sg.level = sg.parent >= 0 ? segs[sg.parent].level + 1 : 0;
        if (sg.parent >= 0) {
          Seg& ps = segs[sg.parent];
          if (ps.nch >= 8) { fits = false; break; }
          ps.ch[ps.nch++] = (int)segs.size();
        }
        seg_of[k] = (int)segs.size();
        segs.push_back(sg);
      }
      if (segs.size() > 255) fits = false;
      if (fits) {
        // elimination steps: the kernel's segment code is scalar in (t, b), so the segments of one step must be alike: a step is
        // a group (tree level, t, b); children sit at a deeper level, i.e. in a later step, and are eliminated first
        std::vector<std::array<int, 3>> groups;
        for (const Seg& sg : segs) {
          std::array<int, 3> k{sg.level, dep[sg.top], dep[sg.bottom]};
          if (std::find(groups.begin(), groups.end(), k) == groups.end()) groups.push_back(k);
        }
        std::sort(groups.begin(), groups.end());
        if (groups.size() > 8) fits = false;
        int mch[8] = {0};
        for (size_t si = 0; si < segs.size() && fits; si++) {
          const Seg& sg = segs[si];
          const int t = dep[sg.top], b = dep[sg.bottom];
          const int step = (int)(std::find(groups.begin(), groups.end(), std::array<int, 3>{sg.level, t, b}) - groups.begin());
          mch[step] = std::max(mch[step], sg.nch);
          uint32_t path[2] = {0, 0}, ch[2] = {0xffffffffu, 0xffffffffu};
          for (int k = sg.bottom; k >= 0; k = dpar[k]) path[dep[k] >> 2] |= (uint32_t)k << (8 * (dep[k] & 3));
          for (int c = 0; c < sg.nch; c++) ch[c >> 2] = (ch[c >> 2] & ~(255u << (8 * (c & 3)))) | ((uint32_t)sg.ch[c] << (8 * (c & 3)));
          int32_t* e = &m->seg_tab[(size_t)sg.top * 6];
          e[0] = t | (b << 4) | (step << 8) | ((int)si << 16);
          e[1] = (int32_t)path[0]; e[2] = (int32_t)path[1]; e[3] = (int32_t)ch[0]; e[4] = (int32_t)ch[1];
        }
        d.seg_lvtb[0] = d.seg_lvtb[1] = 0;
        if (fits) {
          d.seg_nlevel = (int)groups.size();
          for (size_t l = 0; l < groups.size(); l++) {
            d.seg_lvinfo[l >> 2] |= mch[l] << (8 * (l & 3));
            d.seg_lvtb[l >> 2] |= (groups[l][1] | (groups[l][2] << 4)) << (8 * (l & 3));
          }
        }
        m->nseg = (int)segs.size() + 1;   // + the all-zero slot
        d.seg_zero = (int)segs.size();
      }
    }
    if (!fits) { d.seg_nlevel = 0; m->nseg = 0; }
    if (!d.gen && d.nv > 4 && !fits && d.integrator != MM_INT_IMPLICITFAST) d.gen = 1;
  }
  {
    const int32_t* et = (const int32_t*)(blob + m->sec[MM_SEC_EQ_TYPE]);
    for (int e = 0; e < d.neq; e++)
      if (et[e] != MM_EQ_JOINT) { delete m; return fail(MM_EUNSUPPORTED, "only joint equalities are implemented"); }
    const int32_t* gt = (const int32_t*)(blob + m->sec[MM_SEC_GEOM_TYPE]);
    const int32_t* p1 = (const int32_t*)(blob + m->sec[MM_SEC_PAIR_GEOM1]);
    const int32_t* p2 = (const int32_t*)(blob + m->sec[MM_SEC_PAIR_GEOM2]);
    const int32_t* pc = (const int32_t*)(blob + m->sec[MM_SEC_PAIR_CONDIM]);
    for (int p = 0; p < d.npair; p++) {
      const int t1 = gt[p1[p]], t2 = gt[p2[p]];
      const bool ok = (t1 == MM_GEOM_PLANE && (t2 == MM_GEOM_SPHERE || t2 == MM_GEOM_CAPSULE || t2 == MM_GEOM_ELLIPSOID || t2 == MM_GEOM_CYLINDER || t2 == MM_GEOM_BOX)) ||
                      (t1 == MM_GEOM_SPHERE && (t2 == MM_GEOM_SPHERE || t2 == MM_GEOM_CAPSULE || t2 == MM_GEOM_ELLIPSOID || t2 == MM_GEOM_CYLINDER || t2 == MM_GEOM_BOX)) ||
                      (t1 == MM_GEOM_CAPSULE && (t2 == MM_GEOM_CAPSULE || t2 == MM_GEOM_ELLIPSOID || t2 == MM_GEOM_CYLINDER || t2 == MM_GEOM_BOX));
      if (t1 == MM_GEOM_PLANE && (t2 == MM_GEOM_CYLINDER || t2 == MM_GEOM_BOX)) {
        // up to four contacts: two consecutive identical entries, two contacts each (include/myosim_model.h, PAIR_* sections)
        const bool twin = (p > 0 && p1[p - 1] == p1[p] && p2[p - 1] == p2[p]) || (p + 1 < d.npair && p1[p + 1] == p1[p] && p2[p + 1] == p2[p]);
        if (!twin) { delete m; return fail(MM_EBADBLOB, "a plane-box / plane-cylinder pair takes two consecutive entries of the PAIR_* sections"); }
      }
      if (!ok) { delete m; return fail(MM_EUNSUPPORTED, "contact pair types: plane vs sphere/capsule/ellipsoid/cylinder/box, sphere/capsule among themselves, sphere/capsule vs ellipsoid/cylinder/box (geom1 type <= geom2 type)"); }
      if (pc[p] == 4) m->d.condim4 = 1;
      if (pc[p] != 1 && pc[p] != 3 && pc[p] != 4) { delete m; return fail(MM_EUNSUPPORTED, "contact condim must be 1, 3 or 4 (pyramidal cone: 1 / 4 / 6 rows; rolling friction, condim 6, is not implemented)"); }
    }
  }
  if (d.nv > 255) { delete m; return fail(MM_EUNSUPPORTED, "nv > 255"); }
  {
    const int32_t* jlim = (const int32_t*)(blob + m->sec[MM_SEC_JNT_LIMITED]);
    const float* jr = (const float*)(blob + m->sec[MM_SEC_JNT_RANGE]);
    const float* jm = (const float*)(blob + m->sec[MM_SEC_JNT_MARGIN]);
    for (int j = 0; j < d.njnt; j++)
      if (jlim[j] && jr[2 * j + 1] - jr[2 * j] < 2.f * jm[j]) { delete m; return fail(MM_EUNSUPPORTED, "joint range narrower than 2*margin"); }
  }
  const float* damp = (const float*)(blob + m->sec[MM_SEC_DOF_DAMPING]);
  d.any_damping = 0;
  for (int i = 0; i < d.nv; i++) if (damp[i] > 0.f) d.any_damping = 1;
  m->nvp = 0;
  for (int c : kNvpChoices) if (d.nv <= c) { m->nvp = c; break; }
  if (!m->nvp) { delete m; return fail(MM_EUNSUPPORTED, "nv larger than the largest compiled dense tile (40)"); }

  // ---- engine-private tables
  const int32_t* bpar = (const int32_t*)(blob + m->sec[MM_SEC_BODY_PARENT]);
  const int32_t* brootid = (const int32_t*)(blob + m->sec[MM_SEC_BODY_ROOTID]);
  const int32_t* bdofadr = (const int32_t*)(blob + m->sec[MM_SEC_BODY_DOFADR]);
  const int32_t* bdofnum = (const int32_t*)(blob + m->sec[MM_SEC_BODY_DOFNUM]);
  const int32_t* dofbody = (const int32_t*)(blob + m->sec[MM_SEC_DOF_BODYID]);
  std::vector<int32_t> depth(d.nbody, 0), roots, rootslot(d.nbody, 0), dofslot(d.nv, 0);
  for (int b = 1; b < d.nbody; b++) depth[b] = depth[bpar[b]] + 1;
  for (int b = 1; b < d.nbody; b++) if (bpar[b] == 0) roots.push_back(b);
  for (int b = 1; b < d.nbody; b++)
    for (size_t r = 0; r < roots.size(); r++) if (roots[r] == brootid[b]) rootslot[b] = (int)r;
  for (int i = 0; i < d.nv; i++) dofslot[i] = rootslot[dofbody[i]];
  const int32_t* tj_adr = (const int32_t*)(blob + m->sec[MM_SEC_TENJ_ADR]);
  const int32_t* tj_dof = (const int32_t*)(blob + m->sec[MM_SEC_TENJ_DOF]);
  const int32_t* wt = (const int32_t*)(blob + m->sec[MM_SEC_WRAP_TYPE]);
  const int32_t* wo = (const int32_t*)(blob + m->sec[MM_SEC_WRAP_OBJID]);
  const int32_t* tadr = (const int32_t*)(blob + m->sec[MM_SEC_TENDON_ADR]);
  const int32_t* tnum = (const int32_t*)(blob + m->sec[MM_SEC_TENDON_NUM]);
  const int32_t* sbody = (const int32_t*)(blob + m->sec[MM_SEC_SITE_BODYID]);
  const int32_t* gbody = (const int32_t*)(blob + m->sec[MM_SEC_GEOM_BODYID]);
  auto elem_body = [&](int k) -> int {
    if (wt[k] == MM_WRAP_SITE) return sbody[wo[k]];
    if (wt[k] == MM_WRAP_SPHERE || wt[k] == MM_WRAP_CYLINDER) return gbody[wo[k]];
    return -1;
  };
  bool seg_ok = true;
  struct Cross { int ent, ep; };   // J entry a straight segment contributes to, and the end that moves with the dof
  auto crossings = [&](int t, int b0, int b1) {
    // dofs in chain(b0) XOR chain(b1): endpoint 0 for the b0 side (sign -), endpoint 1 for the b1 side (+)
    std::vector<Cross> out;
    while (b0 != b1) {
      int b, ep;
      if (b0 > b1) { b = b0; ep = 0; b0 = bpar[b0]; } else { b = b1; ep = 1; b1 = bpar[b1]; }
      for (int i = bdofadr[b]; i >= 0 && i < bdofadr[b] + bdofnum[b]; i++) {
        int ent = -1;
        for (int e = tj_adr[t]; e < tj_adr[t + 1]; e++) if (tj_dof[e] == i) ent = e;
        if (ent < 0) { seg_ok = false; continue; }
        out.push_back(Cross{ent, ep});
      }
    }
    return out;
  };
  // body frames of the reference configuration (joints at their reference values: the relative pose of two bodies that no dof
  // separates does not depend on the configuration)
  std::vector<double> ref_xp(3 * (size_t)d.nbody, 0.0), ref_xq(4 * (size_t)d.nbody, 0.0);
  {
    const float* bpos = (const float*)(blob + m->sec[MM_SEC_BODY_POS]);
    const float* bquat = (const float*)(blob + m->sec[MM_SEC_BODY_QUAT]);
    ref_xq[0] = 1.0;
    for (int b = 1; b < d.nbody; b++) {
      const double* pq = &ref_xq[4 * bpar[b]];
      const double w = pq[0], x = pq[1], y = pq[2], z = pq[3];
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
      for (int k = 0; k < 3; k++)
        ref_xp[3 * b + k] = ref_xp[3 * bpar[b] + k] + R[3 * k] * bpos[3 * b] + R[3 * k + 1] * bpos[3 * b + 1] + R[3 * k + 2] * bpos[3 * b + 2];
      const double a0 = bquat[4 * b], a1 = bquat[4 * b + 1], a2 = bquat[4 * b + 2], a3 = bquat[4 * b + 3];
      double q[4] = {w * a0 - x * a1 - y * a2 - z * a3, w * a1 + x * a0 + y * a3 - z * a2,
                     w * a2 - x * a3 + y * a0 + z * a1, w * a3 + x * a2 - y * a1 + z * a0};
      const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; k++) ref_xq[4 * b + k] = n > 0 ? q[k] / n : (k == 0);
    }
  }
  m->ten_len0.assign((size_t)std::max(d.ntendon, 1), 0.0);
  m->baked_body.assign((size_t)std::max(d.nbody, 1), 0);
  const float* spos = (const float*)(blob + m->sec[MM_SEC_SITE_POS]);
  // flattened path items (see Engine::tendon): wraps first, then straight segments, then fixed-tendon joint terms
  std::vector<int32_t> item_tab;
  {
    const float* wprm = (const float*)(blob + m->sec[MM_SEC_WRAP_PRM]);
    struct Item { int w[8]; };
    std::vector<Item> wraps, straights, joints;
    auto fbits = [](float f) { int32_t i; memcpy(&i, &f, 4); return i; };
    for (int t = 0; t < d.ntendon; t++) {
      int adr = tadr[t], num = tnum[t], j = 0;
      float inv_div = 1.f;
      for (int k = 0; k < num; k++)
        if (wt[adr + k] == MM_WRAP_JOINT) joints.push_back(Item{{t, 3, adr + k, wo[adr + k], fbits(wprm[adr + k]), 0, 0, fbits(1.f)}});
      while (j < num - 1) {
        int t0 = wt[adr + j], t1 = wt[adr + j + 1];
        if (t0 == MM_WRAP_JOINT) { j++; continue; }
        if (t0 == MM_WRAP_PULLEY || t1 == MM_WRAP_PULLEY) {
          if (t0 == MM_WRAP_PULLEY) inv_div = 1.f / wprm[adr + j];
          j++;
          continue;
        }
        const int k0 = adr + j;
        if (t1 == MM_WRAP_SITE) {
          // A straight segment between two sites whose bodies no dof separates (the same bone, or bones fixed to one another) has
          // the same length in every pose and no Jacobian entry: it is summed into the tendon's constant here -- MyoSuite's muscle
          // paths are mostly such via-point runs along a bone -- instead of being re-measured by a lane in every forward pass.
          if (MM_FOLD_RIGID_SEGMENTS) {
            int b0 = sbody[wo[k0]], b1 = sbody[wo[k0 + 1]];
            std::vector<int> spanned;
            bool rigid = true;
            while (b0 != b1 && rigid) {
              const int b = b0 > b1 ? b0 : b1;
              if (bdofnum[b] > 0) rigid = false;
              spanned.push_back(b);
              if (b0 > b1) b0 = bpar[b0]; else b1 = bpar[b1];
            }
            if (rigid) {
              double p[2][3];
              for (int e = 0; e < 2; e++) {
                const int si = wo[k0 + e], sb = sbody[si];
                const double* q = &ref_xq[4 * sb];
                const double w = q[0], x = q[1], y = q[2], z = q[3];
                const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
                for (int k = 0; k < 3; k++)
                  p[e][k] = ref_xp[3 * sb + k] + R[3 * k] * (double)spos[3 * si] + R[3 * k + 1] * (double)spos[3 * si + 1] + R[3 * k + 2] * (double)spos[3 * si + 2];
              }
              const double dx = p[1][0] - p[0][0], dy = p[1][1] - p[0][1], dz = p[1][2] - p[0][2];
              m->ten_len0[t] += std::sqrt(dx * dx + dy * dy + dz * dz) * (double)inv_div;
              for (int b : spanned) m->baked_body[b] = 1;
              m->nfolded++;
              j += 1;
              continue;
            }
          }
          straights.push_back(Item{{t, 0, k0, wo[k0], wo[k0 + 1], 0, -1, fbits(inv_div)}});
          j += 1;
        } else {
          int side = (int)lrintf(wprm[k0 + 1]);
          wraps.push_back(Item{{t, t1 == MM_WRAP_CYLINDER ? 2 : 1, k0, wo[k0], wo[k0 + 2], wo[k0 + 1], side, fbits(inv_div)}});
          j += 2;
        }
      }
    }
    // spheres and cylinders apart, so that a sweep of lanes runs one wrap flavour
    std::stable_sort(wraps.begin(), wraps.end(), [](const Item& x, const Item& y) { return x.w[1] > y.w[1]; });
    for (auto* v : {&wraps, &straights, &joints})
      for (const Item& it : *v) for (int k = 0; k < 8; k++) item_tab.push_back(it.w[k]);
    m->nwrapitem = (int)wraps.size();
  }
  // Tendon Jacobian by ENTRY (see Engine::tendon): every sparse-J entry (tendon, dof) gets the list of path segments that cross
  // its dof, one 4-word row per segment: S a site-site segment; a wrap item contributes its unwrapped segment A (site - site)
  // or, when the tendon touches the geom, B (site - tangent point) and / or C (tangent point - site): rows A_OR_B, A_OR_C (the
  // usual case: the dof lies between one site's body and the geom's body), B_ONLY, C_ONLY, A_ONLY; J a fixed-tendon coefficient;
  // NONE pads an entry nothing crosses.  Row: [entry | joint word << 16, site0 | site1 << 16, body0 | body1 << 8 | mode << 16 |
  // ep_unwrapped << 20 | ep_wrapped << 21 | wrap slot << 22, bits(1/divisor or coef)]; joint word = joint id | 1 (hinge) or
  // 2 (slide) << 8 -- the kernel reads anchor / axis straight from the joint -- or dof id for ball / free dofs (via cdof).
  // jent[i] = first row | rows << 24 of the i-th entry in processing order.
  std::vector<int32_t> jent, jrow, jent_td;     // jent_td[i] = tendon | dof << 16 of the i-th entry in processing order (the fused tendon-velocity sum)
  {
    enum { R_S = 0, R_AB = 1, R_AC = 2, R_B = 3, R_C = 4, R_A = 5, R_J = 6, R_NONE = 7 };
    struct Rec { int32_t sites, bm, wi, f2; };
    std::vector<std::vector<Rec>> per_ent((size_t)d.ntenJ);
    const int nit = (int)item_tab.size() / 8;
    for (int ii = 0; ii < nit && seg_ok; ii++) {
      const int32_t* I = &item_tab[8 * (size_t)ii];
      const int t = I[0], kind = I[1], k0 = I[2];
      if (kind == 3) {
        const int32_t* jdof = (const int32_t*)(blob + m->sec[MM_SEC_JNT_DOFADR]);
        const int dof = jdof[I[3]];
        int ent = -1;
        for (int e = tj_adr[t]; e < tj_adr[t + 1]; e++) if (tj_dof[e] == dof) ent = e;
        if (ent < 0) { seg_ok = false; break; }
        per_ent[ent].push_back(Rec{0, R_J << 16, 0, I[4]});
        continue;
      }
      if (I[3] >= 65536 || I[4] >= 65536 || sbody[I[3]] >= 256 || sbody[I[4]] >= 256) { seg_ok = false; break; }
      const int32_t sites = I[3] | (I[4] << 16), bodies = sbody[I[3]] | (sbody[I[4]] << 8);
      if (kind == 0) {
        for (const Cross& c : crossings(t, elem_body(k0), elem_body(k0 + 1)))
          per_ent[c.ent].push_back(Rec{sites, bodies | (R_S << 16) | (c.ep << 20), 0, I[7]});
        continue;
      }
      const int b0 = elem_body(k0), b1 = elem_body(k0 + 1), b2 = elem_body(k0 + 2);
      std::vector<Cross> ca = crossings(t, b0, b2), cb = crossings(t, b0, b1), cc = crossings(t, b1, b2);
      auto take = [](std::vector<Cross>& v, int ent, int& ep) {
        for (size_t k = 0; k < v.size(); k++) if (v[k].ent == ent) { ep = v[k].ep; v.erase(v.begin() + k); return true; }
        return false;
      };
      for (const Cross& a_ : ca) {
        int epw = 0;
        if (take(cb, a_.ent, epw)) per_ent[a_.ent].push_back(Rec{sites, bodies | (R_AB << 16) | (a_.ep << 20) | (epw << 21), ii, I[7]});
        else if (take(cc, a_.ent, epw)) per_ent[a_.ent].push_back(Rec{sites, bodies | (R_AC << 16) | (a_.ep << 20) | (epw << 21), ii, I[7]});
        else per_ent[a_.ent].push_back(Rec{sites, bodies | (R_A << 16) | (a_.ep << 20), ii, I[7]});
      }
      for (const Cross& b_ : cb) per_ent[b_.ent].push_back(Rec{sites, bodies | (R_B << 16) | (b_.ep << 21), ii, I[7]});
      for (const Cross& c_ : cc) per_ent[c_.ent].push_back(Rec{sites, bodies | (R_C << 16) | (c_.ep << 21), ii, I[7]});
    }
    if (!seg_ok) { delete m; return fail(MM_EUNSUPPORTED, "tendon Jacobian pattern in the blob does not cover a path segment"); }
    // entries with the most rows first, then by the flavour of their first row: a sweep of lanes runs alike
    std::vector<int> order((size_t)d.ntenJ);
    for (int e = 0; e < d.ntenJ; e++) { order[e] = e; if (per_ent[e].empty()) per_ent[e].push_back(Rec{0, R_NONE << 16, 0, 0}); }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      if (per_ent[x].size() != per_ent[y].size()) return per_ent[x].size() > per_ent[y].size();
      return ((per_ent[x][0].bm >> 16) & 15) > ((per_ent[y][0].bm >> 16) & 15);
    });
    const int32_t* dofjnt = (const int32_t*)(blob + m->sec[MM_SEC_DOF_JNTID]);
    const int32_t* jtype = (const int32_t*)(blob + m->sec[MM_SEC_JNT_TYPE]);
    for (int e : order) {
      const int dof = tj_dof[e], j = dofjnt[dof], ty = jtype[j];
      const bool direct = (ty == MM_JNT_HINGE || ty == MM_JNT_SLIDE) && j < 256;
      if ((!direct && dof >= 256) || e >= 65536 || per_ent[e].size() > 127 || jrow.size() / 4 >= (1u << 24)) {
        delete m; return fail(MM_EUNSUPPORTED, "tendon Jacobian beyond the engine's table limits");
      }
      const int32_t jw = (direct ? j : dof) | ((direct ? (ty == MM_JNT_HINGE ? 1 : 2) : 0) << 8);
      jent.push_back((int32_t)(jrow.size() / 4) | ((int32_t)per_ent[e].size() << 24));
      {
        int te = 0;
        while (te + 1 < d.ntendon && tj_adr[te + 1] <= e) te++;
        if (te >= 65536 || dof >= 65536) { delete m; return fail(MM_EUNSUPPORTED, "tendon Jacobian beyond the engine's table limits"); }
        jent_td.push_back(te | (dof << 16));
      }
      for (const Rec& r : per_ent[e]) {
        if (r.wi >= 1024) { delete m; return fail(MM_EUNSUPPORTED, "more than 1024 wrapping tendon path items"); }
        const int32_t row[4] = {e | (jw << 16), r.sites, r.bm | (r.wi << 22), r.f2};
        for (int k = 0; k < 4; k++) jrow.push_back(row[k]);
      }
    }
  }

  std::vector<uint32_t> dev(m->h_blob);
  auto append = [&](const std::vector<int32_t>& v) {
    int off = (int)dev.size();
    for (int32_t x : v) dev.push_back((uint32_t)x);
    if (v.empty()) dev.push_back(0);
    return off;
  };
  // internal world-frame origin (Dims::ox/oy/oz): mean body position of the reference configuration, on a 1/64 m grid.
  // Bodies hanging off a free joint start wherever qpos0 puts them, which body_pos already encodes.
  {
    const float* bpos = (const float*)(blob + m->sec[MM_SEC_BODY_POS]);
    const float* bquat = (const float*)(blob + m->sec[MM_SEC_BODY_QUAT]);
    std::vector<double> xp(3 * d.nbody, 0.0), xq(4 * d.nbody, 0.0);
    xq[0] = 1.0;
    double sum[3] = {0, 0, 0};
    for (int b = 1; b < d.nbody; b++) {
      const double* pq = &xq[4 * bpar[b]];
      const double w = pq[0], x = pq[1], y = pq[2], z = pq[3];
      const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                           2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                           2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
      for (int k = 0; k < 3; k++)
        xp[3 * b + k] = xp[3 * bpar[b] + k] + R[3 * k] * bpos[3 * b] + R[3 * k + 1] * bpos[3 * b + 1] + R[3 * k + 2] * bpos[3 * b + 2];
      const double a0 = bquat[4 * b], a1 = bquat[4 * b + 1], a2 = bquat[4 * b + 2], a3 = bquat[4 * b + 3];
      double q[4] = {w * a0 - x * a1 - y * a2 - z * a3, w * a1 + x * a0 + y * a3 - z * a2,
                     w * a2 - x * a3 + y * a0 + z * a1, w * a3 + x * a2 - y * a1 + z * a0};
      const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int k = 0; k < 4; k++) xq[4 * b + k] = n > 0 ? q[k] / n : (k == 0);
      for (int k = 0; k < 3; k++) sum[k] += xp[3 * b + k];
    }
    const int nb1 = d.nbody > 1 ? d.nbody - 1 : 1;
    m->origin[0] = (float)(std::round(64.0 * sum[0] / nb1) / 64.0);
    m->origin[1] = (float)(std::round(64.0 * sum[1] / nb1) / 64.0);
    m->origin[2] = (float)(std::round(64.0 * sum[2] / nb1) / 64.0);
    d.ox = m->origin[0]; d.oy = m->origin[1]; d.oz = m->origin[2];
  }
  m->x.body_depth = append(depth); m->x.body_rootslot = append(rootslot); m->x.dof_rootslot = append(dofslot);
  m->x.root_list = append(roots); m->x.nroot = (int)roots.size();
  m->x.jent = append(jent);
  if (MM_FUSE_TENVEL) m->x.jent_td = append(jent_td); else m->x.jent_td = 0;      // (only the fused tendon-velocity sum reads it: no LDS words for an OFF switch)
  while (dev.size() % 4) dev.push_back(0u);   // 16-byte rows
  m->x.jrec = append(jrow);
  {
    // 4-word device rows: [tendon | kind << 16, site0 | site1 << 16 (kind 3: joint id), geom | (sidesite + 1) << 16 (kind 3:
    // bits(coef)), bits(1 / divisor)]
    std::vector<int32_t> packed;
    const int nit = (int)item_tab.size() / 8;
    for (int ii = 0; ii < nit; ii++) {
      const int32_t* I = &item_tab[8 * (size_t)ii];
      if (I[0] >= 65536 || (I[1] != 3 && (I[5] >= 65536 || I[6] + 1 >= 65536))) { delete m; return fail(MM_EUNSUPPORTED, "tendon path beyond the engine's table limits"); }
      packed.push_back(I[0] | (I[1] << 16));
      packed.push_back(I[1] == 3 ? I[3] : (I[3] | (I[4] << 16)));
      packed.push_back(I[1] == 3 ? I[4] : (I[5] | ((I[6] + 1) << 16)));
      packed.push_back(I[7]);
    }
    while (dev.size() % 4) dev.push_back(0u);   // 16-byte rows
    m->x.item_tab = append(packed); m->x.nitem = nit;
    // the tendons' constant length (path segments folded at create): [ntendon] float for the fp32 kernels, then [ntendon] double
    // (8-byte aligned) for the precision-mode kernels
    std::vector<int32_t> l0;
    const int nt_ = std::max(d.ntendon, 1), ntp = (nt_ + 1) & ~1;
    for (int t = 0; t < ntp; t++) { const float f = t < d.ntendon ? (float)m->ten_len0[t] : 0.f; int32_t w; memcpy(&w, &f, 4); l0.push_back(w); }
    for (int t = 0; t < nt_; t++) { const double v = t < d.ntendon ? m->ten_len0[t] : 0.0; int32_t w[2]; memcpy(w, &v, 8); l0.push_back(w[0]); l0.push_back(w[1]); }
    while (dev.size() % 2) dev.push_back(0u);
    m->x.ten_len0 = append(l0); m->x.ten_len0_f64 = m->x.ten_len0 + ntp;
  }
  {
    const int32_t* dpar = (const int32_t*)(blob + m->sec[MM_SEC_DOF_PARENTID]);
    std::vector<int32_t> rel(2 * (size_t)d.nv, 0);
    for (int i = 0; i < d.nv && d.nv <= 64; i++)
      for (int k = i; k >= 0; k = dpar[k]) {   // k is an ancestor-or-self of i: the pair is on one chain, both ways
        rel[2 * i + (k >> 5)] |= (int32_t)(1u << (k & 31));
        rel[2 * k + (i >> 5)] |= (int32_t)(1u << (i & 31));
      }
    m->x.dof_rel = append(rel);
    std::vector<int32_t> bm(2 * (size_t)d.nbody, 0);
    for (int b = 1; b < d.nbody && d.nv <= 64; b++) {
      if (bpar[b] > 0) { bm[2 * b] = bm[2 * bpar[b]]; bm[2 * b + 1] = bm[2 * bpar[b] + 1]; }     // parent < child: already final
      for (int i = bdofadr[b]; i >= 0 && i < bdofadr[b] + bdofnum[b]; i++) bm[2 * b + (i >> 5)] |= (int32_t)(1u << (i & 31));
    }
    m->x.body_dofmask = append(bm);
    m->x.dof_desc = append(m->desc_all); m->x.dof_seg = append(m->seg_tab); m->x.dof_anc = append(m->anc_tab);
    {
      // one word pair per joint for the per-body joint loops (Engine::kinematics / velocity_bias): type | dofadr << 4 | qposadr << 14,
      // bits(qpos0[qposadr])
      const int32_t* jt = (const int32_t*)(blob + m->sec[MM_SEC_JNT_TYPE]);
      const int32_t* jd = (const int32_t*)(blob + m->sec[MM_SEC_JNT_DOFADR]);
      const int32_t* jq = (const int32_t*)(blob + m->sec[MM_SEC_JNT_QPOSADR]);
      const uint32_t* q0 = blob + m->sec[MM_SEC_QPOS0];
      std::vector<int32_t> jp(2 * (size_t)std::max(d.njnt, 1), 0);
      for (int j = 0; j < d.njnt; j++) {
        if (jd[j] < 0 || jd[j] >= 1024 || jq[j] < 0 || jq[j] >= 1024) { delete m; return fail(MM_EUNSUPPORTED, "joint addresses beyond the engine's packed joint word (1024 dofs / qpos words)"); }
        jp[2 * j] = (int32_t)(jt[j] | (jd[j] << 4) | (jq[j] << 14));
        jp[2 * j + 1] = (int32_t)q0[jq[j]];
      }
      m->x.jnt_pack = append(jp);
    }
    {
      // chains of the body tree (Engine::subtree_sum).  A body starts a chain when it hangs off the world or its parent has
      // another child too; the bodies of a chain must have consecutive ids (MuJoCo's depth-first numbering gives that).
      std::vector<int32_t> tab(3 * (size_t)std::max(d.nbody, 1), -1);
      std::vector<int> nchb(d.nbody, 0), top_of(d.nbody, 0), lvl(d.nbody, 0);
      for (int b = 1; b < d.nbody; b++) if (bpar[b] > 0) nchb[bpar[b]]++;
      bool ok = d.nbody <= 255;
      int nlev = 0;
      for (int b = 1; b < d.nbody && ok; b++) {
        const int p = bpar[b];
        if (p > 0 && nchb[p] == 1) {             // continues its parent's chain
          if (p != b - 1) { ok = false; break; }
          top_of[b] = top_of[p];
          tab[3 * (size_t)top_of[b]] = (tab[3 * (size_t)top_of[b]] & ~255) | b;   // new bottom
          continue;
        }
        top_of[b] = b;
        lvl[b] = p > 0 ? lvl[top_of[p]] + 1 : 0;
        if (lvl[b] > 15) { ok = false; break; }
        nlev = std::max(nlev, lvl[b] + 1);
        tab[3 * (size_t)b] = b | (lvl[b] << 8);
        tab[3 * (size_t)b + 1] = 0; tab[3 * (size_t)b + 2] = 0;
        if (p > 0) {                             // register with the chain it hangs off (whose bottom is p)
          int32_t* pt = &tab[3 * (size_t)top_of[p]];
          const int c = (pt[0] >> 12) & 15;
          if (c >= 8) { ok = false; break; }
          pt[1 + (c >> 2)] |= (int32_t)((uint32_t)b << (8 * (c & 3)));
          pt[0] = (pt[0] & ~(15 << 12)) | ((c + 1) << 12);
        }
      }
      int maxch = 0, maxlen = 1;
      for (int b = 1; b < d.nbody && ok; b++)
        if (tab[3 * (size_t)b] >= 0) { maxch = std::max(maxch, (tab[3 * (size_t)b] >> 12) & 15); maxlen = std::max(maxlen, (tab[3 * (size_t)b] & 255) - b + 1); }
      d.bchain_nlevel = ok ? (nlev | (maxch << 4) | (maxlen << 8)) : 0;
      m->x.body_chain = append(tab);
    }
  }
  m->blob_words = (int)dev.size();
  m->cofs = (int)dev.size();          // ConstBlock (dims / LDS layout / aux offsets): global-only tail, not staged into LDS
  dev.resize(dev.size() + (sizeof(ConstBlock) + 3) / 4, 0u);
  m->cofs_tw = (int)dev.size();       // the same for two-wave launches (their LDS layout differs)
  dev.resize(dev.size() + (sizeof(ConstBlock) + 3) / 4, 0u);

  // default group width: the smallest that can own every body / dof / constraint row and has a compiled kernel
  m->lanes = 0;
  const int rk4 = integ_kernel(d.integrator);
  for (int c : {4, 8, 16, 32, 64}) if (check_lanes(m, c) && have_kernel(c, m->nvp, d.gen, rk4)) { m->lanes = c; break; }
  if (!m->lanes) {
    // a model whose rows need a wider group than its dofs do (torso: 18 dofs, 33 rows): take the next larger dense tile
    // that has a kernel at that width (the padding dofs are inert)
    const int nvp_min = m->nvp;
    for (int c : {4, 8, 16, 32, 64}) {
      for (int n : kNvpChoices) {
        if (n <= nvp_min) continue;
        m->nvp = n;
        if (check_lanes(m, c) && have_kernel(c, n, d.gen, rk4)) { m->lanes = c; break; }
      }
      if (m->lanes) break;
    }
    if (!m->lanes) m->nvp = nvp_min;
  }
  if (!m->lanes) { delete m; return fail(MM_EUNSUPPORTED, "no compiled kernel owns this model (needs > 64 lanes per env: nbody, nv, njnt or constraint rows > 64)"); }
  if (d.gen || rk4) m->lanes_auto = 0;   // row tables are sized for one group width; RK4 kernels exist for the default width only
  build_layout(m);
  HIPCHK(hipGetDevice(&m->device));
  HIPCHK(hipMalloc((void**)&m->d_blob, dev.size() * sizeof(uint32_t)));
  HIPCHK(hipMemcpy(m->d_blob, dev.data(), dev.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  { int rc = upload_consts(m); if (rc != MM_OK) return rc; }
  *out = m;
  return MM_OK;
}

// dims / LDS layout / aux offsets as the kernel reads them (ConstBlock behind the blob); re-sent when the layout changes
static int upload_consts(mm_model* m) {
  ConstBlock cb;
  memset(&cb, 0, sizeof(cb));
  cb.d = m->d; cb.L = m->L; cb.x = m->x;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(m->d_blob + m->cofs, &cb, sizeof(cb), hipMemcpyHostToDevice));
  cb.L = m->Ltw;   // the const block of two-wave launches: the same dims and tables, the other LDS layout
  HIPCHK(hipMemcpy(m->d_blob + m->cofs_tw, &cb, sizeof(cb), hipMemcpyHostToDevice));
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
  if (!check_lanes(m, lanes) || !have_model_kernel(m, lanes))
    return fail(MM_EARG, "lanes_per_env must be 4/8/16/32/64, >= nbody, nv, njnt, padded nv (and constraint rows), with a compiled kernel");
  m->lanes = lanes;
  m->lanes_auto = 0;
  m->lanes_user = 1;
  build_layout(m);
  return upload_consts(m);
}

extern "C" int mm_model_set_option(mm_model* m, const char* name, int value) {
  if (!m || !name) return MM_EARG;
  if (!strcmp(name, "lds_model")) { m->lds_model = value; return MM_OK; }
  if (!strcmp(name, "waves_per_block")) { m->waves_per_block = value; return MM_OK; }
  if (!strcmp(name, "precision")) {
    // MM_PREC_F32 (default): the fp32 kernels.  MM_PREC_F64: fp64 arithmetic, registers and LDS tables; state rows stay fp32 (a
    // drop-in for every caller).  MM_PREC_F64_STATE: the four state rows of mm_state are fp64 as well.  (include/myosim.h)
    if (value != MM_PREC_F32 && value != MM_PREC_F64 && value != MM_PREC_F64_STATE) return fail(MM_EARG, "precision: MM_PREC_F32 / MM_PREC_F64 / MM_PREC_F64_STATE");
    if (value != MM_PREC_F32) {
      bool any = false;
      for (int c : {4, 8, 16, 32, 64}) any = any || (check_lanes(m, c) && have_kernel_f64(c, m->nvp, m->d.gen, integ_kernel(m->d.integrator)));
      if (!any) return fail(MM_EUNSUPPORTED, "precision: no fp64 kernel for this model (compiled: limit-rows-only models with nv <= 24 on Euler; general-row models with nv <= 36 at 64 lanes per env on Euler, 36-wide also implicitfast; no RK4)");
    }
    const int old = m->precision;
    m->precision = value;
    if (m->lanes_user && !have_model_kernel(m, m->lanes)) { m->precision = old; return fail(MM_EUNSUPPORTED, "precision: no kernel of that family at the pinned lanes_per_env"); }
    if (!m->lanes_user && !have_model_kernel(m, m->lanes)) {   // default width of the family (a general-row model's default is fixed, not pinned: the fp64 general-row kernels are 64 lanes wide)
      for (int c : {64, 32, 16, 8, 4}) if (check_lanes(m, c) && have_model_kernel(m, c)) m->lanes = c;
    }
    build_layout(m);
    return upload_consts(m);
  }
  // mjOption.iterations / ls_iterations of THIS model handle (the blob's values are the default): the reference's MJX envs overwrite
  // them after loading the model (envs/myo/mjx/mjx_base_env.py:50-51: spec.option.iterations = 6, ls_iterations = 6)
  if (!strcmp(name, "iterations") || !strcmp(name, "ls_iterations")) {
    if (value < 1 || value > 1000) return fail(MM_EARG, "iterations / ls_iterations: 1 ... 1000");
    if (name[0] == 'i') m->d.iterations = value; else m->d.ls_iterations = value;
    return upload_consts(m);
  }
  if (!strcmp(name, "origin_shift")) {   // 0: the kernel works in raw world coordinates (A/B of the fp32 error study)
    m->d.ox = value ? m->origin[0] : 0.f; m->d.oy = value ? m->origin[1] : 0.f; m->d.oz = value ? m->origin[2] : 0.f;
    return upload_consts(m);
  }
  return fail(MM_EARG, "unknown option");
}

static bool have_obs_kernel(int G, int nvp, int gen, int rk4);
// mm_task.fwd_carry: the fp32 Euler kernels of 8 dofs and more (Engine::CARRY), and only where the action reaches nothing but act_dot
// -- every actuator has activation dynamics
static bool fwd_carry_ok(const mm_model* m) {
  if (m->d.integrator == MM_INT_RK4 || m->precision != MM_PREC_F32 || m->d.nu == 0 || m->nvp < 8) return false;
  const int32_t* dt = (const int32_t*)(m->h_blob.data() + m->sec[MM_SEC_ACT_DYNTYPE]);
  for (int u = 0; u < m->d.nu; u++) if (dt[u] == MM_DYN_NONE) return false;
  return true;
}
extern "C" int mm_model_info(const mm_model* m, int which) {
  if (!m) return MM_EARG;
  switch (which) {
    case MM_INFO_NQ: return m->d.nq; case MM_INFO_NV: return m->d.nv; case MM_INFO_NU: return m->d.nu;
    case MM_INFO_NA: return m->d.na; case MM_INFO_NBODY: return m->d.nbody; case MM_INFO_NSITE: return m->d.nsite;
    case MM_INFO_NTENDON: return m->d.ntendon; case MM_INFO_LANES_PER_ENV: return m->lanes;
    case MM_INFO_LDS_BYTES_PER_ENV: return (int)m->lds_per_env;
    case MM_INFO_ENVS_PER_BLOCK: return (64 / m->lanes) * (m->waves_per_block > 0 ? m->waves_per_block : 1);
    case MM_INFO_NGEOM: return m->d.ngeom; case MM_INFO_WAVES_PER_BLOCK: return m->waves_per_block;
    case MM_INFO_MODEL_WORDS: return m->blob_words;
    case MM_INFO_BODY_CHAINS: return m->d.bchain_nlevel;
    case MM_INFO_FOLDED_RESET: return (MM_FOLD_RESET && m->lanes == 64 && !m->lanes_auto && have_obs_kernel(64, m->nvp, m->d.gen, integ_kernel(m->d.integrator))) ? 1 : 0;
    case MM_INFO_FWD_CARRY: return fwd_carry_ok(m) ? 1 : 0;
    case MM_INFO_TENDON_ITEMS: return m->x.nitem;
    case MM_INFO_TENDON_FOLDED: return m->nfolded;
    case MM_INFO_KERNEL_FAMILY: return m->d.gen ? 2 : ((MM_SPARSE_LDL && m->nvp >= 8 && m->d.integrator != MM_INT_IMPLICITFAST) ? 1 : 0);
  }
  return MM_EARG;
}

// debug-record layout query (tests): offset of a named field in the per-env dump record
extern "C" int mm_debug_layout(const mm_model* m, const char* name) {
  const DbgLayout& D = m->D;
#define LQ(n) if (!strcmp(name, #n)) return D.n;
  LQ(xpos) LQ(xquat) LQ(xipos) LQ(cdof) LQ(cvel) LQ(tenlen) LQ(tenvel) LQ(tenj) LQ(actfrc) LQ(actdot) LQ(M) LQ(bias)
  LQ(smooth) LQ(qaccsm) LQ(qacc) LQ(qfrccon) LQ(efc_active) LQ(efc_D) LQ(efc_aref) LQ(scal) LQ(total)
#undef LQ
  return -1;
}

static float* g_dbg = nullptr;
extern "C" void mm_debug_set_dump(float* dev_ptr) { g_dbg = dev_ptr; }
static unsigned long long* g_prof = nullptr;
extern "C" void mm_debug_set_prof(unsigned long long* dev_ptr) { g_prof = dev_ptr; }

// mm_model_launch_info: when set, the launch path stops short of the launch and reports the geometry and the kernel's
// occupancy instead (bench.py prices counters per RESIDENT wave with it)
struct LaunchInfo { int* out; };
static thread_local LaunchInfo* g_info = nullptr;
static int report_kernel(const void* fn, dim3 grid, dim3 block, size_t lds, int lanes, int two_wave, int lm) {
  int nb = 0;
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, (int)block.x, lds));
  hipFuncAttributes fa;
  HIPCHK(hipFuncGetAttributes(&fa, fn));
  int* o = g_info->out;
  o[MM_LAUNCH_LANES] = lanes; o[MM_LAUNCH_WAVES_PER_BLOCK] = (int)block.x / 64; o[MM_LAUNCH_TWO_WAVE] = two_wave;
  o[MM_LAUNCH_LDS_MODEL] = lm; o[MM_LAUNCH_LDS_BYTES] = (int)lds; o[MM_LAUNCH_BLOCKS] = (int)grid.x;
  o[MM_LAUNCH_RESIDENT_BLOCKS_PER_CU] = nb; o[MM_LAUNCH_VGPRS] = fa.numRegs;
  return MM_OK;
}

template <int G, int NVP, bool GEN, int RK4>
static int launch_t(const mm_model* m, KArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st, int lm) {
  // the dynamic-LDS limit is a per-device attribute of the function: one flag per (device, LM variant) of this instantiation
  static std::atomic<unsigned> attr_done[2] = {{0u}, {0u}};   // bit d = set on device d (devices >= 32: set on every launch)
  const unsigned bit = m->device < 32 ? (1u << m->device) : 0u;
  if (!(attr_done[lm].load(std::memory_order_acquire) & bit) || !bit) {
    if (lm) HIPCHK(hipFuncSetAttribute((const void*)k_engine<G, NVP, true, GEN, RK4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else HIPCHK(hipFuncSetAttribute((const void*)k_engine<G, NVP, false, GEN, RK4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[lm].fetch_or(bit, std::memory_order_release);
  }
  if (g_info) return report_kernel(lm ? (const void*)k_engine<G, NVP, true, GEN, RK4> : (const void*)k_engine<G, NVP, false, GEN, RK4>, grid, block, lds, G, a.two_wave, lm);
  if (lm) hipLaunchKernelGGL((k_engine<G, NVP, true, GEN, RK4>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((k_engine<G, NVP, false, GEN, RK4>), grid, block, lds, st, a);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

template <int G, int NVP, bool GEN, int RK4>
static int launch_f64_t(const mm_model* m, KArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st, int lm) {
  static std::atomic<unsigned> attr_done[2] = {{0u}, {0u}};
  const unsigned bit = m->device < 32 ? (1u << m->device) : 0u;
  if (!(attr_done[lm].load(std::memory_order_acquire) & bit) || !bit) {
    if (lm) HIPCHK(hipFuncSetAttribute((const void*)mm64::k_engine<G, NVP, true, GEN, RK4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else HIPCHK(hipFuncSetAttribute((const void*)mm64::k_engine<G, NVP, false, GEN, RK4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[lm].fetch_or(bit, std::memory_order_release);
  }
  if (g_info) return report_kernel(lm ? (const void*)mm64::k_engine<G, NVP, true, GEN, RK4> : (const void*)mm64::k_engine<G, NVP, false, GEN, RK4>, grid, block, lds, G, a.two_wave, lm);
  if (lm) hipLaunchKernelGGL((mm64::k_engine<G, NVP, true, GEN, RK4>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((mm64::k_engine<G, NVP, false, GEN, RK4>), grid, block, lds, st, a);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

static bool have_obs_kernel(int G, int nvp, int gen, int rk4) {
#define X(G_, N_, GN_, RK_) if (G == G_ && nvp == N_ && gen == GN_ && rk4 == RK_) return true;
  MM_KERNELS_OBS(X)
#undef X
  return false;
}
template <int G, int NVP, bool GEN, int RK4>
static int launch_obs_t(const mm_model* m, KArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  static std::atomic<unsigned> attr_done{0u};
  const unsigned bit = m->device < 32 ? (1u << m->device) : 0u;
  if (!(attr_done.load(std::memory_order_acquire) & bit) || !bit) {
    HIPCHK(hipFuncSetAttribute((const void*)k_engine<G, NVP, false, GEN, RK4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.fetch_or(bit, std::memory_order_release);
  }
  if (g_info) return report_kernel((const void*)k_engine<G, NVP, false, GEN, RK4, true>, grid, block, lds, G, a.two_wave, 0);
  hipLaunchKernelGGL((k_engine<G, NVP, false, GEN, RK4, true>), grid, block, lds, st, a);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

// group width (lanes per env) a launch over `nenv` envs uses: the pinned / default width, or -- for models without general
// constraint rows, whose LDS tables do not depend on the width -- the narrowest group (most envs per wave) that still yields
// >= 2 waves per CU, else the widest available
static int pick_lanes(const mm_model* m, int nenv) {
  int G = m->lanes;
  if (m->lanes_auto && !m->d.gen) {
    int best = 0;
    for (int c : {4, 8, 16, 32, 64}) {
      if (!check_lanes(m, c) || !have_model_kernel(m, c)) continue;
      best = c;
      if ((nenv + (64 / c) - 1) / (64 / c) >= 512) break;
    }
    if (best) G = best;
  }
  return G;
}

extern "C" int mm_model_launch_lanes(const mm_model* m, int nenv) {
  if (!m || nenv <= 0) return MM_EARG;
  return pick_lanes(m, nenv);
}

static int launch_on_device(const mm_model* m, KArgs& a, void* stream, const int G);
static int launch(const mm_model* m, KArgs& a, void* stream);
static void fill_common(const mm_model* m, KArgs& a, const mm_state* s);
// geometry and occupancy of the env-step launch over `nenv` envs (include/myosim.h: MM_LAUNCH_*); nothing is launched
extern "C" int mm_model_launch_info(const mm_model* m, int nenv, int* out, int nout) {
  if (!m || nenv <= 0 || !out || nout < MM_LAUNCH_COUNT) return fail(MM_EARG, "mm_model_launch_info: bad argument");
  mm_state s; memset(&s, 0, sizeof(s)); s.nenv = nenv; s.geom_env_id = -1;
  KArgs a; fill_common(m, a, &s);
  a.mode = 2;
  LaunchInfo li{out};
  g_info = &li;
  const int rc = launch(m, a, nullptr);
  g_info = nullptr;
  return rc;
}
static int launch(const mm_model* m, KArgs& a, void* stream) {
  if (a.s.body_pos_env && a.s.body_pos_env_id > 0 && a.s.body_pos_env_id < (int)m->baked_body.size() && m->baked_body[a.s.body_pos_env_id])
    return fail(MM_EUNSUPPORTED, "mm_state.body_pos_env names a body whose frame position is part of a tendon path segment folded into the tendon's "
                                 "constant length at mm_model_create (a segment between two bones that no dof separates)");
  const int G = pick_lanes(m, a.s.nenv);
  {   // launch on the model's device (the caller's stream must belong to it); restore the caller's current device afterwards
    int cur = -1;
    HIPCHK(hipGetDevice(&cur));
    if (cur != m->device) {
      HIPCHK(hipSetDevice(m->device));
      KArgs& a2 = a;
      const int rc = launch_on_device(m, a2, stream, G);
      (void)hipSetDevice(cur);
      return rc;
    }
  }
  return launch_on_device(m, a, stream, G);
}

static int launch_on_device(const mm_model* m, KArgs& a, void* stream, const int G) {
  const int epw = 64 / G;
  const size_t kLds = 160 * 1024;
  const size_t blob_bytes = (size_t)((m->blob_words + 3) & ~3) * 4;
  const int waves_needed = (a.s.nenv + epw - 1) / epw;
  // lds_model: 1 = stage the model tables in LDS unless that costs resident waves the batch needs (then read them through
  // L2 instead: a graceful step instead of an occupancy cliff when a model grows past the LDS budget), 0 = never, 2 = always
  int want = (waves_needed + 255) / 256;       // waves per CU that spread the batch over all 256 CUs in one round
  if (want < 1) want = 1;
  if (want > 8) want = 8;
  // Two waves per env group (Engine::TW): the Euler and implicitfast kernels, when the batch leaves at least half of the
  // SIMDs without a wave (<= 4 env waves per CU: the block still fits the 512-thread launch bound with the helpers in it) and
  // the larger per-env tables (a second dense tile) do not cost env waves
  int two_wave = (g_two_wave && integ_kernel(m->d.integrator) != 1 && want <= 4 && m->waves_per_block <= 0) ? 1 : 0;
  // precision-mode kernels: a lane's register state doubles, so they are built for one wave per SIMD (256-thread blocks, up to
  // 512 VGPRs + AGPRs per lane); no helper waves
  const bool f64 = m->precision != MM_PREC_F32;
  const int max_wpb = f64 ? 4 : 8;              // __launch_bounds__ of the family
  if (f64) { two_wave = 0; if (want > max_wpb) want = max_wpb; }
  // the reset-observation pass of a task (mm_task.obs_only) has its own kernel symbol where one is compiled (model through L2)
  const bool obs_kernel = a.mode == 2 && a.t.obs_only && have_obs_kernel(G, m->nvp, m->d.gen, integ_kernel(m->d.integrator));
  int lm = 0, wpb = 0;
  size_t per_env = 0, model_bytes = 0;
  for (;;) {
    per_env = two_wave ? m->lds_per_env_tw : m->lds_per_env;
    auto fit_waves = [&](size_t mbytes) {   // waves of one block that fit in LDS next to the model copy (<= 8)
      int fit = max_wpb;                       // __launch_bounds__ (512 threads; 256 in precision mode)
      while (fit > 1 && mbytes + (size_t)fit * epw * per_env > kLds) fit--;
      return fit;
    };
    lm = (m->lds_model && !obs_kernel) ? 1 : 0;
    if (m->lds_model == 1 && fit_waves(blob_bytes) < want && fit_waves(0) > fit_waves(blob_bytes)) lm = 0;
    if (m->lds_model == 1 && blob_bytes + (size_t)epw * per_env > kLds) lm = 0;   // not even one wave fits next to the model copy
    model_bytes = lm ? blob_bytes : 0;
    wpb = std::min(m->waves_per_block, max_wpb);
    if (wpb <= 0) {
      // one block per CU sharing one model copy: as many waves as fit in LDS, but no fatter than needed
      wpb = want;
      const int fit = fit_waves(model_bytes);
      if (wpb > fit) wpb = fit;
    }
    if (two_wave && (wpb < want || wpb > 4)) { two_wave = 0; continue; }   // the extra tile would cost env waves: one wave per env
    break;
  }
  const int epb = epw * wpb;
  const size_t lds = model_bytes + (size_t)epb * per_env;
  if (lds > kLds) return fail(MM_ELDS, "per-block LDS tables exceed 160 KiB");
  a.two_wave = two_wave;
  if (two_wave) { a.cofs = m->cofs_tw; a.L = m->Ltw; }
  dim3 grid((a.s.nenv + epb - 1) / epb), block(64 * wpb * (a.two_wave ? 2 : 1));
  hipStream_t st = (hipStream_t)stream;
  a.blob_words = m->blob_words;
  a.prof = g_prof;
  const int rk4 = integ_kernel(m->d.integrator);
  a.state_f64 = m->precision == MM_PREC_F64_STATE ? 1 : 0;
  if (f64) {
#define X(G_, N_, GN_, RK_) \
    if (G == G_ && m->nvp == N_ && m->d.gen == GN_ && rk4 == RK_) return launch_f64_t<G_, N_, GN_ != 0, RK_>(m, a, grid, block, lds, st, lm);
    MM_KERNELS_F64(X)
#undef X
    return fail(MM_EUNSUPPORTED, "no compiled precision-mode kernel for this (lanes_per_env, nv) combination");
  }
  if (obs_kernel) {
#define X(G_, N_, GN_, RK_) \
    if (G == G_ && m->nvp == N_ && m->d.gen == GN_ && rk4 == RK_) return launch_obs_t<G_, N_, GN_ != 0, RK_>(m, a, grid, block, lds, st);
    MM_KERNELS_OBS(X)
#undef X
  }
#define X(G_, N_, GN_, RK_) \
  if (G == G_ && m->nvp == N_ && m->d.gen == GN_ && rk4 == RK_) return launch_t<G_, N_, GN_ != 0, RK_>(m, a, grid, block, lds, st, lm);
  MM_KERNEL_LIST(X)
#undef X
  return fail(MM_EUNSUPPORTED, "no compiled kernel for this (lanes_per_env, nv) combination");
}

static void fill_common(const mm_model* m, KArgs& a, const mm_state* s) {
  memset(&a, 0, sizeof(a));
  a.blob = m->d_blob; a.cofs = m->cofs;
  memcpy(a.sec, m->sec, sizeof(a.sec));
  a.d = m->d; a.L = m->L; a.D = m->D; a.x = m->x; a.s = *s;
  if (!a.s.geom_size_env || a.s.geom_env_id < 0 || a.s.geom_env_id >= m->d.ngeom) { a.s.geom_size_env = nullptr; a.s.geom_type_env = nullptr; a.s.geom_env_id = -1; }
  if (!a.s.body_mass_env || a.s.body_mass_env_id <= 0 || a.s.body_mass_env_id >= m->d.nbody) { a.s.body_mass_env = nullptr; a.s.body_mass_env_id = -1; }
  if (!a.s.body_pos_env || a.s.body_pos_env_id <= 0 || a.s.body_pos_env_id >= m->d.nbody) { a.s.body_pos_env = nullptr; a.s.body_pos_env_id = -1; }
}

extern "C" int mm_step(const mm_model* m, const mm_state* s, const float* ctrl, int nsub, void* stream) {
  if (!m || !s || s->nenv <= 0 || nsub < 0) return fail(MM_EARG, "mm_step: bad argument");
  KArgs a; fill_common(m, a, s);
  a.ctrl = ctrl; a.mode = 0; a.t.nsubsteps = nsub; a.dbg = nullptr;
  return launch(m, a, stream);
}

extern "C" int mm_forward(const mm_model* m, const mm_state* s, const float* ctrl, const mm_derived* out, void* stream) {
  if (!m || !s || s->nenv <= 0) return fail(MM_EARG, "mm_forward: bad argument");
  KArgs a; fill_common(m, a, s);
  a.ctrl = ctrl; a.mode = 1;
  if (out) { a.o = *out; a.has_derived = 1; }
  a.dbg = g_dbg;
  return launch(m, a, stream);
}

// mm_task / mm_rollout grow by appending fields: take min(caller's size, ours) bytes, zero the rest (include/myosim.h)
// min_size = the struct as ABI 4 introduced `size` (everything up to mm_task.obs_only / mm_rollout.reset_seed): a caller from
// before that has no size field -- its first word is something else -- and must not slip through on a small value
template <typename T>
static int sized_copy(T* dst, const T* src, size_t min_size, const char* what) {
  if (!src) return fail(MM_EARG, what);
  const uint32_t sz = *reinterpret_cast<const uint32_t*>(src);
  if (sz < min_size || sz > sizeof(T)) return fail(MM_EARG, "mm_task / mm_rollout: .size is unset, smaller than the ABI-4 struct, or larger than this library's struct (caller built against a newer header)");
  memset(dst, 0, sizeof(T));
  memcpy(dst, src, sz);
  dst->size = (uint32_t)sizeof(T);
  return MM_OK;
}

static int check_task(const mm_model* m, const mm_state* s, const mm_task* t) {
  if (!m || !s || s->nenv <= 0 || !t) return fail(MM_EARG, "mm_env_step: bad argument");
  if (t->task == MM_TASK_POSE && !t->target_jnt_value) return fail(MM_EARG, "pose task needs target_jnt_value");
  if (t->fwd_carry && !fwd_carry_ok(m)) return fail(MM_EUNSUPPORTED, "mm_task.fwd_carry: Euler / implicitfast, fp32, nv >= 5, every actuator with activation dynamics (MM_INFO_FWD_CARRY)");
  if (t->task == MM_TASK_REACH && (!t->tip_sites || !t->target_pos || t->ntip <= 0)) return fail(MM_EARG, "reach task needs tip_sites/target_pos");
  if (t->task == MM_TASK_WALK) {
    if (!t->do_forward && !t->obs_only) return fail(MM_EARG, "walk task needs do_forward");
    for (int k = 0; k < 4; k++) if (t->walk_body[k] <= 0 || t->walk_body[k] >= m->d.nbody) return fail(MM_EARG, "walk task: bad body id");
    for (int k = 0; k < 6; k++) if (t->walk_qadr[k] < 0 || t->walk_qadr[k] >= m->d.nq) return fail(MM_EARG, "walk task: bad qpos address");
    if (m->d.nq < 7 || t->walk_hip_period <= 0) return fail(MM_EARG, "walk task needs a free root joint and hip_period > 0");
  }
  if (t->task == MM_TASK_REORIENT) {
    if (!t->do_forward && !t->obs_only) return fail(MM_EARG, "reorient task needs do_forward");
    if (t->reor_obj_body <= 0 || t->reor_obj_body >= m->d.nbody || t->reor_eps_site < 0 || t->reor_eps_site >= m->d.nsite ||
        !t->reor_axis_half || !t->reor_des_rot || !(t->reor_pen_length > 0.f) || m->d.nq < 7)
      return fail(MM_EARG, "reorient task: bad body/site id or missing per-env buffers");
  }
  if (t->task == MM_TASK_OBJHOLD) {
    if (!t->do_forward && !t->obs_only) return fail(MM_EARG, "object-hold task needs do_forward");
    if (!t->tip_sites || !t->target_pos || m->d.nq < 8) return fail(MM_EARG, "object-hold task needs tip_sites[0], target_pos and a free-joint object");
  }
  if (t->task == MM_TASK_KEYTURN) {
    if (!t->do_forward && !t->obs_only) return fail(MM_EARG, "key-turn task needs do_forward");
    if (!t->tip_sites || t->ntip != 3 || m->d.nq < 2) return fail(MM_EARG, "key-turn task needs tip_sites = {keyhead, IFtip, THtip}");
  }
  if (t->task != MM_TASK_NONE && t->task != MM_TASK_POSE && t->task != MM_TASK_REACH && t->task != MM_TASK_WALK &&
      t->task != MM_TASK_REORIENT && t->task != MM_TASK_OBJHOLD && t->task != MM_TASK_KEYTURN)
    return fail(MM_EUNSUPPORTED, "task not implemented");
  if (t->fatigue && (!t->fat_MA || !t->fat_MR || !t->fat_MF)) return fail(MM_EARG, "fatigue needs MA/MR/MF");
  return MM_OK;
}

extern "C" int mm_env_step(const mm_model* m, const mm_state* s, const float* action, const mm_task* t,
                           const mm_derived* out, void* stream) {
  mm_task tt;
  { const int rc = sized_copy(&tt, t, offsetof(mm_task, obs_only) + sizeof(int), "mm_env_step: null task"); if (rc != MM_OK) return rc; }
  t = &tt;
  { const int rc = check_task(m, s, t); if (rc != MM_OK) return rc; }
  KArgs a; fill_common(m, a, s);
  a.ctrl = action; a.mode = 2; a.t = *t;
  if (out) { a.o = *out; a.has_derived = 1; }
  a.dbg = g_dbg;
  return launch(m, a, stream);
}

extern "C" int mm_rollout_step(const mm_model* m, const mm_state* s, const mm_task* t, const mm_rollout* r,
                               const mm_derived* out, void* stream) {
  mm_task tt; mm_rollout rr;
  { const int rc = sized_copy(&tt, t, offsetof(mm_task, obs_only) + sizeof(int), "mm_rollout_step: null task"); if (rc != MM_OK) return rc; }
  { const int rc = sized_copy(&rr, r, offsetof(mm_rollout, reset_seed) + sizeof(uint64_t), "mm_rollout_step: null rollout description"); if (rc != MM_OK) return rc; }
  t = &tt; r = &rr;
  { const int rc = check_task(m, s, t); if (rc != MM_OK) return rc; }
  if (!r) return fail(MM_EARG, "mm_rollout_step: null rollout description");
  if (t->obs_only) return fail(MM_EARG, "mm_rollout_step: obs_only passes go through mm_env_step");
  if (r->autoreset) {
    if (t->task == MM_TASK_POSE) {
      if (!r->tlo || !r->thi || !r->target || !r->episode || !t->step_count || (r->random_qpos && (!r->qlo || !r->qhi)))
        return fail(MM_EARG, "mm_rollout_step: autoreset needs tlo/thi/target/episode/step_count (and qlo/qhi for random_qpos)");
      if (r->target != t->target_jnt_value) return fail(MM_EARG, "mm_rollout_step: rollout.target must be the task's target_jnt_value buffer");
    } else if (t->task == MM_TASK_WALK || t->task == MM_TASK_REORIENT) {
      // the second (reset-observation) pass is decided per wavefront: one env per wave, general-row kernels
      if (!(pick_lanes(m, s->nenv) == 64 && have_obs_kernel(64, m->nvp, m->d.gen, integ_kernel(m->d.integrator))))
        return fail(MM_EUNSUPPORTED, "mm_rollout_step: the folded walk / reorient reset exists in the 64-lane kernels of MM_KERNELS_OBS (reset through reset_mask instead)");
      if (!r->episode || !t->step_count) return fail(MM_EARG, "mm_rollout_step: autoreset needs episode / step_count");
      if (t->task == MM_TASK_WALK && (!r->walk_ka_qpos || !r->walk_ka_qvel || (r->walk_random && (!r->walk_kb_qpos || !r->walk_kb_qvel))))
        return fail(MM_EARG, "mm_rollout_step: walk autoreset needs the key pose(s)");
      if (t->task == MM_TASK_REORIENT && (!r->reor_init_qpos || !r->reor_size_tables || r->reor_ntab <= 0 || !r->reor_geom_size_env || !r->reor_geom_type_env ||
                                          !r->reor_axis_half || !r->reor_des_rot || !(r->reor_tar_length > 0.f) || r->reor_geom_size_env != s->geom_size_env ||
                                          r->reor_geom_type_env != s->geom_type_env || r->reor_axis_half != t->reor_axis_half || r->reor_des_rot != t->reor_des_rot))
        return fail(MM_EARG, "mm_rollout_step: reorient autoreset needs init_qpos / size tables and the state's / task's per-env buffers");
      if (t->fatigue && (!t->fat_MA || !t->fat_MR || !t->fat_MF)) return fail(MM_EARG, "mm_rollout_step: fatigue state missing");
    } else return fail(MM_EUNSUPPORTED, "mm_rollout_step: the folded auto-reset exists for the POSE, WALK and REORIENT tasks (reset the others through reset_mask)");
  }
  KArgs a; fill_common(m, a, s);
  a.ctrl = r->action; a.mode = 2; a.t = *t; a.ro = *r; a.has_ro = 1;
  if (out) { a.o = *out; a.has_derived = 1; }
  a.dbg = g_dbg;
  return launch(m, a, stream);
}

extern "C" int mm_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qpos_src,
                        const float* qvel_src, void* stream) {
  if (!m || !s || s->nenv <= 0) return fail(MM_EARG, "mm_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.qpos_src = qpos_src; r.qvel_src = qvel_src;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_pose_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* qlo,
                             const float* qhi, const float* tlo, const float* thi, float* target, int32_t* episode,
                             int32_t* step_count, uint64_t seed, int random_qpos, float* obs, int obs_dim,
                             int obs_layout, void* stream) {
  if (!m || !s || s->nenv <= 0 || !tlo || !thi || !target) return fail(MM_EARG, "mm_pose_reset: bad argument");
  if (random_qpos && (!qlo || !qhi)) return fail(MM_EARG, "mm_pose_reset: random_qpos needs qlo/qhi");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask;
  r.qlo = qlo; r.qhi = qhi; r.tlo = tlo; r.thi = thi; r.target = target; r.episode = episode;
  r.step_count = step_count; r.seed = seed; r.pose = 1; r.random_qpos = random_qpos;
  r.obs = obs; r.obs_dim = obs_dim; r.obs_layout = obs_layout;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_uniform_at(float* out, size_t n, uint64_t seed, uint64_t stream_id, size_t first_index, void* stream) {
  if (!out) return fail(MM_EARG, "mm_uniform: null output");
  if (n == 0) return MM_OK;
  const size_t ncounter = (first_index + n + 3) / 4 - first_index / 4;
  hipLaunchKernelGGL(k_uniform, dim3((unsigned)((ncounter + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n, seed, stream_id,
                     first_index);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_uniform(float* out, size_t n, uint64_t seed, uint64_t stream_id, void* stream) {
  return mm_uniform_at(out, n, seed, stream_id, 0, stream);
}

// Generalised advantage estimation over an unroll of T steps, one thread per env: brax.training.agents.ppo.losses.compute_gae (the
// learner of benchmarks/mjx_benchmark_PPO.py:50-60), term by term --
//   delta_t = (r_t + gamma (1 - term_t) V_{t+1} - V_t) (1 - trunc_t)          a truncated step does NOT bootstrap from V_{t+1} (under
//                                                                              auto-reset that is the value of the NEXT episode's first observation)
//   acc_t   = delta_t + gamma (1 - term_t)(1 - trunc_t) lambda acc_{t+1}
//   vs_t    = acc_t + V_t                                                      -> returns (the value target)
//   adv_t   = (r_t + gamma (1 - term_t) vs_{t+1} - V_t)(1 - trunc_t),  vs_T = V_T (the bootstrap value)
// Arrays are [T][n] (value [T+1][n]): coalesced over envs.
__global__ void k_gae(const float* rew, const float* term, const float* trunc, const float* val, float* adv, float* ret, int T, int n,
                      float gamma, float lam) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float acc = 0.f;
  float vnext = val[(size_t)T * n + e], vs_next = vnext;
  for (int t = T - 1; t >= 0; t--) {
    const size_t k = (size_t)t * n + e;
    const float nt = 1.f - term[k], tm = 1.f - (trunc ? trunc[k] : 0.f), v = val[k], r = rew[k];
    const float delta = (r + gamma * nt * vnext - v) * tm;
    acc = delta + gamma * nt * tm * lam * acc;
    const float vs = acc + v;
    adv[k] = (r + gamma * nt * vs_next - v) * tm;
    ret[k] = vs;
    vs_next = vs; vnext = v;
  }
}
extern "C" int mm_gae(const float* reward, const float* terminated, const float* truncated, const float* value, float* advantage,
                      float* returns, int T, int nenv, float gamma, float lam, void* stream) {
  if (!reward || !terminated || !value || !advantage || !returns || T <= 0 || nenv <= 0) return fail(MM_EARG, "mm_gae: bad argument");
  hipLaunchKernelGGL(k_gae, dim3((nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, reward, terminated, truncated, value, advantage,
                     returns, T, nenv, gamma, lam);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

__global__ void k_episode_stats(float* stats, uint8_t* mask, const float* rwd, int cols, int dense_col, int solved_col,
                                const uint8_t* done, const uint8_t* trunc, int nenv) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nenv) return;
  const float* r = rwd + (size_t)e * cols;
  float* s = stats + (size_t)e * 3;
  s[0] += r[dense_col]; s[1] += 1.f; s[2] = fmaxf(s[2], r[solved_col]);
  if (mask) mask[e] = (uint8_t)((done && done[e]) || (trunc && trunc[e]));
}

extern "C" int mm_episode_stats(float* stats, uint8_t* reset_mask, const float* rwd, int rwd_cols, int dense_col, int solved_col,
                                const uint8_t* done, const uint8_t* truncated, int nenv, void* stream) {
  if (!stats || !rwd || nenv <= 0 || dense_col < 0 || dense_col >= rwd_cols || solved_col < 0 || solved_col >= rwd_cols)
    return fail(MM_EARG, "mm_episode_stats: bad argument");
  hipLaunchKernelGGL(k_episode_stats, dim3((nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, stats, reset_mask, rwd, rwd_cols,
                     dense_col, solved_col, done, truncated, nenv);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

// 3CC-r fatigue state of the masked envs back to rest (fatigue.py:82-99): MF = vec (or 0), MR = 1 - MF, MA = 0
__global__ void k_fatigue_reset(float* MA, float* MR, float* MF, const uint8_t* mask, const float* vec, int nenv, int na) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nenv * na) return;
  const int e = i / na, k = i - e * na;
  if (mask && !mask[e]) return;
  const float f = vec ? vec[k] : 0.f;
  MA[i] = 0.f; MR[i] = 1.f - f; MF[i] = f;
}

extern "C" int mm_fatigue_reset(float* MA, float* MR, float* MF, const uint8_t* mask, const float* vec, int nenv, int na,
                                void* stream) {
  if (!MA || !MR || !MF || nenv <= 0 || na <= 0) return fail(MM_EARG, "mm_fatigue_reset: bad argument");
  hipLaunchKernelGGL(k_fatigue_reset, dim3((nenv * na + 255) / 256), dim3(256), 0, (hipStream_t)stream, MA, MR, MF, mask, vec, nenv, na);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

__global__ void k_env_draw(float* out, int nenv, int ncomp, const float* base, const float* lo, const float* hi,
                           const uint8_t* mask, const int32_t* episode, uint64_t seed, uint32_t stream_id, int env_index_base) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nenv || (mask && !mask[e])) return;
  const uint32_t ep = episode ? (uint32_t)episode[e] : 0u;
  for (int k = 0; k < ncomp; k++) {
    uint32_t c[4] = {(uint32_t)(k >> 2), stream_id, (uint32_t)(env_index_base + e), ep};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    out[(size_t)e * ncomp + k] = (base ? base[k] : 0.f) + lo[k] + (hi[k] - lo[k]) * u01(c[k & 3]);
  }
}

extern "C" int mm_env_draw(float* out, int nenv, int ncomp, const float* base, const float* lo, const float* hi,
                           const uint8_t* mask, const int32_t* episode, uint64_t seed, uint32_t stream_id, int env_index_base,
                           void* stream) {
  if (!out || !lo || !hi || nenv <= 0 || ncomp <= 0) return fail(MM_EARG, "mm_env_draw: bad argument");
  hipLaunchKernelGGL(k_env_draw, dim3((nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, out, nenv, ncomp, base, lo, hi,
                     mask, episode, seed, stream_id, env_index_base);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_reach_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* tlo, const float* thi,
                              float* target, const float* tip0, int ntip, int32_t* episode, int32_t* step_count,
                              uint64_t seed, float* obs, int obs_dim, void* stream) {
  if (!m || !s || s->nenv <= 0 || !tlo || !thi || !target || !tip0 || ntip <= 0) return fail(MM_EARG, "mm_reach_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask;
  r.tlo = tlo; r.thi = thi; r.target = target; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.reach = 1; r.ntip = ntip; r.tip0 = tip0; r.obs = obs; r.obs_dim = obs_dim;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_walk_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* key_a_qpos,
                             const float* key_a_qvel, const float* key_b_qpos, const float* key_b_qvel, int random,
                             int32_t* episode, int32_t* step_count, uint64_t seed, void* stream) {
  if (!m || !s || s->nenv <= 0 || !key_a_qpos || !key_a_qvel) return fail(MM_EARG, "mm_walk_reset: bad argument");
  if (random && (!key_b_qpos || !key_b_qvel)) return fail(MM_EARG, "mm_walk_reset: random reset needs the second key");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.walk = 1; r.walk_random = random; r.ka_qpos = key_a_qpos; r.ka_qvel = key_a_qvel; r.kb_qpos = key_b_qpos; r.kb_qvel = key_b_qvel;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_reorient_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                                 const float* size_table, int ntab, float* geom_size_env, float* axis_half, float* des_rot,
                                 float tar_length, int32_t* episode, int32_t* step_count, uint64_t seed, void* stream) {
  if (!m || !s || s->nenv <= 0 || !init_qpos || !size_table || ntab <= 0 || !geom_size_env || !axis_half || !des_rot || !(tar_length > 0.f))
    return fail(MM_EARG, "mm_reorient_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.qpos_bcast = init_qpos;
  r.reor = 1; r.reor_ntab = ntab; r.reor_tab = size_table; r.reor_gsize = geom_size_env; r.reor_axis_half = axis_half;
  r.reor_des_rot = des_rot; r.reor_tar_length = tar_length;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_reorient_reset_typed(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                                       const float* size_tables, int ntab, float* geom_size_env, int32_t* geom_type_env,
                                       float* axis_half, float* des_rot, float tar_length, int32_t* episode,
                                       int32_t* step_count, uint64_t seed, void* stream) {
  if (!m || !s || s->nenv <= 0 || !init_qpos || !size_tables || ntab <= 0 || !geom_size_env || !geom_type_env || !axis_half || !des_rot ||
      !(tar_length > 0.f))
    return fail(MM_EARG, "mm_reorient_reset_typed: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.qpos_bcast = init_qpos;
  r.reor = 1; r.reor_ntab = ntab; r.reor_tab = size_tables; r.reor_gsize = geom_size_env; r.reor_axis_half = axis_half;
  r.reor_des_rot = des_rot; r.reor_tar_length = tar_length; r.reor_gtype = geom_type_env;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_pen_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos, float axis_half,
                            float lo0, float hi0, float lo1, float hi1, float* des_rot, float tar_length, int32_t* episode,
                            int32_t* step_count, uint64_t seed, void* stream) {
  if (!m || !s || s->nenv <= 0 || !init_qpos || !des_rot || !(tar_length > 0.f) || !(axis_half > 0.f))
    return fail(MM_EARG, "mm_pen_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.qpos_bcast = init_qpos;
  r.reor = 1; r.pen = 1; r.pen_axis_half = axis_half; r.pen_lo0 = lo0; r.pen_hi0 = hi0; r.pen_lo1 = lo1; r.pen_hi1 = hi1;
  r.reor_des_rot = des_rot; r.reor_tar_length = tar_length;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}

extern "C" int mm_objhold_reset(const mm_model* m, const mm_state* s, const uint8_t* mask, const float* init_qpos,
                                const float* goal_center, float goal_half, float size_lo, float size_hi, float* goal,
                                float* geom_size_env, int32_t* episode, int32_t* step_count, uint64_t seed, void* stream) {
  if (!m || !s || s->nenv <= 0 || !init_qpos || !goal_center || !goal) return fail(MM_EARG, "mm_objhold_reset: bad argument");
  ResetArgs r; memset(&r, 0, sizeof(r));
  r.blob = m->d_blob; r.state_f64 = m->precision == MM_PREC_F64_STATE; r.qpos0_off = m->sec[MM_SEC_QPOS0]; r.nq = m->d.nq; r.nv = m->d.nv; r.na = m->d.na;
  r.nenv = s->nenv; r.s = *s; r.mask = mask; r.episode = episode; r.step_count = step_count; r.seed = seed;
  r.qpos_bcast = init_qpos;
  r.hold = 1; r.hold_center = goal_center; r.hold_half = goal_half; r.hold_slo = size_lo; r.hold_shi = size_hi;
  r.hold_goal = goal; r.hold_gsize = geom_size_env;
  hipLaunchKernelGGL(k_reset, dim3((s->nenv + 255) / 256), dim3(256), 0, (hipStream_t)stream, r);
  HIPCHK(hipGetLastError());
  return MM_OK;
}
