/* mmo_engine.c -- fp64 CPU ORACLE for the batched musculoskeletal physics step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (myosuite_amd/) may
 * import, link or call this file; it exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the HIP engine against an
 * independent double-precision restatement of the same algorithm.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in the
 * third-party MuJoCo C library (`mujoco>=3.6,<3.7`, /root/reference/pyproject.toml:31;
 * lock file pins mujoco 3.5.0, uv.lock:1739-1740), which is absent from
 * /root/reference and from this image, as are the myo_sim model files.  The
 * reference holds no golden vectors for mj_step.  This file
 * therefore restates MuJoCo's *published* per-step pipeline (documentation
 * chapters "Computation" and "Modeling/Muscles", mjModel/mjData field semantics)
 * in the stage order of mj_step, anchored on the reference's call sites:
 *   mj_step loop .......... myosuite/robot/robot.py:856-861
 *   mj_forward ............ myosuite/robot/robot.py:595-607 (sensor2sim)
 *   mj_resetData .......... myosuite/robot/robot.py:999-1002
 *   muscle activation ..... tutorials/6_Inverse_Dynamics.ipynb:231-237,288-291
 * Stage map:
 *   A1 kinematics+comPos  -> mmo_kinematics, mmo_com_pos
 *   A2 spatial tendons    -> mmo_tendon (+ mmo_wrap)
 *   A3 transmission       -> mmo_transmission
 *   A4 CRB + L'DL         -> mmo_crb, mmo_factor
 *   A5 velocity stage     -> mmo_com_vel, mmo_passive, mmo_rne
 *   A6 muscle actuation   -> mmo_actuation
 *   A7 constraints        -> mmo_make_constraint, mmo_solve (Newton)
 *   A9 integration        -> mmo_euler
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/myosim_model.h"

#define MINVAL MM_MINVAL
#ifndef MMO_REAL_EXTERNAL   /* tests/tools/count_flops.py compiles this file as C++ with `real` = an operation-counting class */
#ifndef MMO_REAL
#define MMO_REAL double   /* -DMMO_REAL=float builds the fp32 rounding-error study variant */
#endif
typedef MMO_REAL real;
#endif

/* stage markers (no-ops here): tests/tools/precision_study.py compiles this file with a scalar type that rounds to fp32 inside
   the stages it selects, to see which stage's precision the 1000-step divergence hangs on */
#ifndef MMO_STAGE
#define MMO_STAGE(k)
#endif
enum { MMO_ST_KIN = 0, MMO_ST_COM, MMO_ST_TENDON, MMO_ST_CRB, MMO_ST_CONSTR, MMO_ST_VEL, MMO_ST_ACT, MMO_ST_ACC, MMO_ST_SOLVE, MMO_ST_INTEG, MMO_ST_NONE };

/* ------------------------------------------------------------------ model */
typedef struct {
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, ntendon, nwrap, neq, npair, nM, njmax, nconmax;
  int iterations, ls_iterations, integrator, eulerdamp;
  real timestep, gravity[3], tolerance, ls_tolerance, meaninertia, impratio;
  const int32_t* I[MM_NSEC];  /* integer sections (NULL for float sections)   */
  real* F[MM_NSEC];           /* float sections upcast to double              */
  int len[MM_NSEC];
  uint32_t* blob;
} mmo_model;

#define MI(m, S) ((m)->I[MM_SEC_##S])
#define MF(m, S) ((m)->F[MM_SEC_##S])

static const char kSecType[MM_NSEC] = {
#define X(NAME, T, W) T,
    MM_SECTIONS(X)
#undef X
};

mmo_model* mmo_model_load(const uint32_t* blob, int nwords) {
  if (nwords < MM_HEADER_WORDS || blob[0] != MM_MAGIC || blob[1] != MM_VERSION ||
      blob[2] != MM_NSEC || (int)blob[3] != nwords)
    return NULL;
  mmo_model* m = (mmo_model*)calloc(1, sizeof(mmo_model));
  m->blob = (uint32_t*)malloc(sizeof(uint32_t) * nwords);
  memcpy(m->blob, blob, sizeof(uint32_t) * nwords);
  for (int s = 0; s < MM_NSEC; s++) {
    int off = (int)m->blob[MM_HEADER_WORDS + 2 * s], n = (int)m->blob[MM_HEADER_WORDS + 2 * s + 1];
    m->len[s] = n;
    if (kSecType[s] == 'i') {
      m->I[s] = (const int32_t*)(m->blob + off);
    } else {
      m->F[s] = (real*)malloc(sizeof(real) * (n > 0 ? n : 1));
      const float* src = (const float*)(m->blob + off);
      for (int k = 0; k < n; k++) m->F[s][k] = (real)src[k];
    }
  }
  const int32_t* oi = MI(m, OPT_I);
  const real* of = MF(m, OPT_F);
  m->nq = oi[MM_OI_NQ]; m->nv = oi[MM_OI_NV]; m->nu = oi[MM_OI_NU]; m->na = oi[MM_OI_NA];
  m->nbody = oi[MM_OI_NBODY]; m->njnt = oi[MM_OI_NJNT]; m->ngeom = oi[MM_OI_NGEOM];
  m->nsite = oi[MM_OI_NSITE]; m->ntendon = oi[MM_OI_NTENDON]; m->nwrap = oi[MM_OI_NWRAP];
  m->neq = oi[MM_OI_NEQ]; m->npair = oi[MM_OI_NPAIR]; m->nM = oi[MM_OI_NM];
  m->njmax = oi[MM_OI_NJMAX]; m->nconmax = oi[MM_OI_NCONMAX];
  m->iterations = oi[MM_OI_ITERATIONS]; m->ls_iterations = oi[MM_OI_LS_ITERATIONS];
  m->integrator = oi[MM_OI_INTEGRATOR]; m->eulerdamp = oi[MM_OI_EULERDAMP];
  m->timestep = of[MM_OF_TIMESTEP];
  m->gravity[0] = of[MM_OF_GRAV_X]; m->gravity[1] = of[MM_OF_GRAV_Y]; m->gravity[2] = of[MM_OF_GRAV_Z];
  m->tolerance = of[MM_OF_TOLERANCE]; m->ls_tolerance = of[MM_OF_LS_TOLERANCE];
  m->meaninertia = of[MM_OF_MEANINERTIA]; m->impratio = of[MM_OF_IMPRATIO];
  return m;
}

void mmo_model_free(mmo_model* m) {
  if (!m) return;
  for (int s = 0; s < MM_NSEC; s++) free(m->F[s]);
  free(m->blob);
  free(m);
}

/* ------------------------------------------------------------------- data */
typedef struct {
  /* state */
  real time, *qpos, *qvel, *act, *ctrl, *qacc_warmstart;
  /* position stage */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *site_xpos, *geom_xpos, *geom_xmat;
  real *subtree_com, *cinert, *cdof, *crb;
  real *ten_length, *ten_J, *actuator_length, *actuator_moment;
  real *qM, *qLD, *qLDiagInv;
  /* velocity stage */
  real *ten_velocity, *actuator_velocity, *cvel, *cdof_dot, *qfrc_passive, *qfrc_bias;
  /* actuation / acceleration */
  real *act_dot, *actuator_force, *qfrc_actuator, *qfrc_smooth, *qacc_smooth;
  /* constraints */
  int nefc, ncon;
  int col_seq, col_part;   /* mmo_collide: contact number within the pair entry being collided, the entry's part (mmo_collision.inc) */
  int *efc_type, *efc_id;
  real *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_solref, *efc_solimp;
  real* efc_floss;   /* friction-loss rows: the dry-friction bound (0 on every other row) */
  real *efc_R, *efc_D, *efc_vel, *efc_aref, *efc_force;
  real *qfrc_constraint, *qacc;
  /* per-env model delta: size override of one geom (mm_state.geom_size_env) */
  int gsize_id, gtype; real gsize_val[3];
  /* per-env model deltas on one body each (mm_state.body_mass_env / body_pos_env) */
  int bmass_id, bpos_id; real bmass_val, bpos_val[3];
  /* contacts */
  int* con_pair;
  real *con_dist, *con_pos, *con_frame;
  /* scratch */
  real *cacc, *cfrc, *tmp_nv, *qH, *qHDiagInv;
  /* scratch stack for the temporaries of the step path (salloc / srelease): no heap traffic per call, so that many
     threads stepping their own mmo_data never meet in the allocator (bench.py's cpu_baseline leg) */
  real* scratch; int scratch_cap, scratch_top;
  /* fp32-state twin (tests): round qpos / qvel / act / qacc_warmstart to float after every mmo_step */
  int round_state_f32;
  /* per-user task parameters the engine does not interpret */
  int solver_niter, warn_bad;
  /* profiling counters */
  double flops;
} mmo_data;

static real* ralloc(int n) { return (real*)calloc(n > 0 ? n : 1, sizeof(real)); }
/* zeroed temporaries from the data's scratch stack; release in LIFO order with the mark taken before the first salloc */
static real* salloc(mmo_data* d, int n) {
  if (n < 1) n = 1;
  if (d->scratch_top + n > d->scratch_cap) { fprintf(stderr, "mmo: scratch stack overflow (%d + %d > %d)\n", d->scratch_top, n, d->scratch_cap); abort(); }
  real* p = d->scratch + d->scratch_top;
  d->scratch_top += n;
  memset((void*)p, 0, sizeof(real) * (size_t)n);
  return p;
}
static int smark(const mmo_data* d) { return d->scratch_top; }
static void srelease(mmo_data* d, int mark) { d->scratch_top = mark; }

void mmo_reset(const mmo_model* m, mmo_data* d);

mmo_data* mmo_data_create(const mmo_model* m) {
  mmo_data* d = (mmo_data*)calloc(1, sizeof(mmo_data));
  int nv = m->nv, nb = m->nbody, nt = m->ntendon, nu = m->nu, nj = m->njmax > 0 ? m->njmax : 1;
  d->qpos = ralloc(m->nq); d->qvel = ralloc(nv); d->act = ralloc(m->na); d->ctrl = ralloc(nu);
  d->qacc_warmstart = ralloc(nv);
  d->xpos = ralloc(3 * nb); d->xquat = ralloc(4 * nb); d->xmat = ralloc(9 * nb);
  d->xipos = ralloc(3 * nb); d->ximat = ralloc(9 * nb);
  d->xanchor = ralloc(3 * m->njnt); d->xaxis = ralloc(3 * m->njnt);
  d->site_xpos = ralloc(3 * m->nsite); d->geom_xpos = ralloc(3 * m->ngeom); d->geom_xmat = ralloc(9 * m->ngeom);
  d->subtree_com = ralloc(3 * nb); d->cinert = ralloc(10 * nb); d->cdof = ralloc(6 * nv); d->crb = ralloc(10 * nb);
  d->ten_length = ralloc(nt); d->ten_J = ralloc(nt * nv);
  d->actuator_length = ralloc(nu); d->actuator_moment = ralloc(nu * nv);
  d->qM = ralloc(m->nM); d->qLD = ralloc(m->nM); d->qLDiagInv = ralloc(nv);
  d->ten_velocity = ralloc(nt); d->actuator_velocity = ralloc(nu);
  d->cvel = ralloc(6 * nb); d->cdof_dot = ralloc(6 * nv);
  d->qfrc_passive = ralloc(nv); d->qfrc_bias = ralloc(nv);
  d->act_dot = ralloc(m->na); d->actuator_force = ralloc(nu); d->qfrc_actuator = ralloc(nv);
  d->qfrc_smooth = ralloc(nv); d->qacc_smooth = ralloc(nv);
  d->efc_type = (int*)calloc(nj, sizeof(int)); d->efc_id = (int*)calloc(nj, sizeof(int));
  d->efc_J = ralloc(nj * nv); d->efc_pos = ralloc(nj); d->efc_margin = ralloc(nj);
  d->efc_diagApprox = ralloc(nj); d->efc_solref = ralloc(2 * nj); d->efc_solimp = ralloc(5 * nj);
  d->efc_R = ralloc(nj); d->efc_D = ralloc(nj); d->efc_vel = ralloc(nj); d->efc_aref = ralloc(nj);
  d->efc_force = ralloc(nj); d->efc_floss = ralloc(nj);
  d->qfrc_constraint = ralloc(nv); d->qacc = ralloc(nv);
  {
    int nc = m->nconmax > 0 ? m->nconmax : 1;
    d->con_pair = (int*)calloc(nc, sizeof(int)); d->con_dist = ralloc(nc); d->con_pos = ralloc(3 * nc);
    d->con_frame = ralloc(9 * nc);
  }
  d->cacc = ralloc(6 * nb); d->cfrc = ralloc(6 * nb); d->tmp_nv = ralloc(nv);
  d->qH = ralloc(m->nM); d->qHDiagInv = ralloc(nv);
  /* deepest nesting: mmo_rk4 (nq + nv + 5 na + 8 nv) -> forward -> mmo_solve (5 nv + 3 njmax + nv^2) or mmo_implicitfast (2 nv^2);
     mmo_collide_and_add 9 nv, mmo_tendon 6 nv, mmo_com_pos nbody */
  d->scratch_cap = m->nq + 34 * nv + 5 * m->na + 4 * nj + 2 * nv * nv + nb + 64;   /* (+ 10 nv: the torsional rows of one condim-4 contact) */
  d->scratch = ralloc(d->scratch_cap); d->scratch_top = 0;
  d->gsize_id = -1; d->gtype = -1; d->bmass_id = -1; d->bpos_id = -1;
  mmo_reset(m, d);
  return d;
}

void mmo_data_free(mmo_data* d) {
  if (!d) return;
  real** p[] = {&d->qpos, &d->qvel, &d->act, &d->ctrl, &d->qacc_warmstart, &d->xpos, &d->xquat, &d->xmat,
                &d->xipos, &d->ximat, &d->xanchor, &d->xaxis, &d->site_xpos, &d->geom_xpos, &d->geom_xmat,
                &d->subtree_com, &d->cinert, &d->cdof, &d->crb, &d->ten_length, &d->ten_J,
                &d->actuator_length, &d->actuator_moment, &d->qM, &d->qLD, &d->qLDiagInv, &d->ten_velocity,
                &d->actuator_velocity, &d->cvel, &d->cdof_dot, &d->qfrc_passive, &d->qfrc_bias, &d->act_dot,
                &d->actuator_force, &d->qfrc_actuator, &d->qfrc_smooth, &d->qacc_smooth, &d->efc_J,
                &d->efc_pos, &d->efc_margin, &d->efc_diagApprox, &d->efc_solref, &d->efc_solimp, &d->efc_R,
                &d->efc_D, &d->efc_vel, &d->efc_aref, &d->efc_force, &d->qfrc_constraint, &d->qacc, &d->cacc,
                &d->cfrc, &d->tmp_nv, &d->qH, &d->qHDiagInv, &d->con_dist, &d->con_pos, &d->con_frame, &d->efc_floss};
  for (unsigned i = 0; i < sizeof(p) / sizeof(p[0]); i++) free(*p[i]);
  free(d->efc_type); free(d->efc_id); free(d->con_pair); free(d->scratch);
  free(d);
}

/* mj_resetData semantics (call site robot.py:999): qpos=qpos0, everything else 0 */
void mmo_reset(const mmo_model* m, mmo_data* d) {
  for (int i = 0; i < m->nq; i++) d->qpos[i] = MF(m, QPOS0)[i];
  memset(d->qvel, 0, sizeof(real) * m->nv);
  memset(d->act, 0, sizeof(real) * m->na);
  memset(d->ctrl, 0, sizeof(real) * m->nu);
  memset(d->qacc_warmstart, 0, sizeof(real) * m->nv);
  d->time = 0;
  d->warn_bad = 0;
}

/* ---------------------------------------------------------- small algebra */
static inline real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real* r, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline real norm3(const real* a) { return sqrt(dot3(a, a)); }
static inline real normalize3(real* a) {
  real n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
  else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static inline void quat_mul(real* r, const real* a, const real* b) {
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_normalize(real* q) {
  real n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static inline void quat2mat(real* m, const real* q) { /* row-major 3x3 */
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[4] = w * w - x * x + y * y - z * z; m[8] = w * w - x * x - y * y + z * z;
  m[1] = 2 * (x * y - w * z); m[3] = 2 * (x * y + w * z);
  m[2] = 2 * (x * z + w * y); m[6] = 2 * (x * z - w * y);
  m[5] = 2 * (y * z - w * x); m[7] = 2 * (y * z + w * x);
}
static inline void mat_vec(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void matT_vec(real* r, const real* m, const real* v) {
  real x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2];
  real y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2];
  real z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void axisangle2quat(real* q, const real* axis, real angle) {
  real s = sin(0.5 * angle);
  q[0] = cos(0.5 * angle); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial inertia (10 numbers: Ixx Iyy Izz Ixy Ixz Iyz, m*r (3), m) times motion vector [w; v] */
static inline void inert_mul(real* res, const real* I, const real* v) {
  const real* w = v; const real* l = v + 3; const real* mr = I + 6;
  real c[3];
  res[0] = I[0] * w[0] + I[3] * w[1] + I[4] * w[2];
  res[1] = I[3] * w[0] + I[1] * w[1] + I[5] * w[2];
  res[2] = I[4] * w[0] + I[5] * w[1] + I[2] * w[2];
  cross3(c, mr, l);
  res[0] += c[0]; res[1] += c[1]; res[2] += c[2];
  cross3(c, mr, w);
  res[3] = I[9] * l[0] - c[0]; res[4] = I[9] * l[1] - c[1]; res[5] = I[9] * l[2] - c[2];
}
static inline void cross_motion(real* res, const real* v, const real* s) {
  real a[3], b[3], c[3];
  cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
  res[0] = a[0]; res[1] = a[1]; res[2] = a[2];
  res[3] = b[0] + c[0]; res[4] = b[1] + c[1]; res[5] = b[2] + c[2];
}
static inline void cross_force(real* res, const real* v, const real* f) {
  real a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  res[0] = a[0] + b[0]; res[1] = a[1] + b[1]; res[2] = a[2] + b[2];
  res[3] = c[0]; res[4] = c[1]; res[5] = c[2];
}

/* --------------------------------------------------------- A1 kinematics */
static void mmo_kinematics(const mmo_model* m, mmo_data* d) {
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  memcpy(d->ximat, d->xmat, 9 * sizeof(real));
  d->xipos[0] = d->xipos[1] = d->xipos[2] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int p = MI(m, BODY_PARENT)[b];
    real pos[3], quat[4], v[3];
    mat_vec(v, d->xmat + 9 * p, b == d->bpos_id ? d->bpos_val : MF(m, BODY_POS) + 3 * b);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + v[k];
    quat_mul(quat, d->xquat + 4 * p, MF(m, BODY_QUAT) + 4 * b);
    int ja = MI(m, BODY_JNTADR)[b], jn = MI(m, BODY_JNTNUM)[b];
    for (int j = ja; j < ja + jn; j++) {
      int type = MI(m, JNT_TYPE)[j], qa = MI(m, JNT_QPOSADR)[j];
      real mat[9];
      if (type == MM_JNT_FREE) {
        for (int k = 0; k < 3; k++) pos[k] = d->qpos[qa + k];
        for (int k = 0; k < 4; k++) quat[k] = d->qpos[qa + 3 + k];
        quat_normalize(quat);
        for (int k = 0; k < 3; k++) d->xanchor[3 * j + k] = pos[k];
        quat2mat(mat, quat);
        d->xaxis[3 * j] = mat[2]; d->xaxis[3 * j + 1] = mat[5]; d->xaxis[3 * j + 2] = mat[8];
        continue;
      }
      quat2mat(mat, quat);
      real anchor[3], axis[3];
      mat_vec(v, mat, MF(m, JNT_POS) + 3 * j);
      for (int k = 0; k < 3; k++) anchor[k] = pos[k] + v[k];
      mat_vec(axis, mat, MF(m, JNT_AXIS) + 3 * j);
      for (int k = 0; k < 3; k++) { d->xanchor[3 * j + k] = anchor[k]; d->xaxis[3 * j + k] = axis[k]; }
      if (type == MM_JNT_SLIDE) {
        real dq = d->qpos[qa] - MF(m, QPOS0)[qa];
        for (int k = 0; k < 3; k++) pos[k] += axis[k] * dq;
      } else if (type == MM_JNT_HINGE) {
        real ql[4], qn[4];
        axisangle2quat(ql, MF(m, JNT_AXIS) + 3 * j, d->qpos[qa] - MF(m, QPOS0)[qa]);
        quat_mul(qn, quat, ql);
        memcpy(quat, qn, sizeof(qn));
        quat2mat(mat, quat);
        mat_vec(v, mat, MF(m, JNT_POS) + 3 * j);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - v[k];
      } else { /* ball */
        real ql[4] = {d->qpos[qa], d->qpos[qa + 1], d->qpos[qa + 2], d->qpos[qa + 3]}, qn[4];
        quat_normalize(ql);
        quat_mul(qn, quat, ql);
        memcpy(quat, qn, sizeof(qn));
        quat2mat(mat, quat);
        mat_vec(v, mat, MF(m, JNT_POS) + 3 * j);
        for (int k = 0; k < 3; k++) pos[k] = anchor[k] - v[k];
      }
    }
    quat_normalize(quat);
    memcpy(d->xpos + 3 * b, pos, sizeof(pos));
    memcpy(d->xquat + 4 * b, quat, sizeof(quat));
    quat2mat(d->xmat + 9 * b, quat);
    mat_vec(v, d->xmat + 9 * b, MF(m, BODY_IPOS) + 3 * b);
    for (int k = 0; k < 3; k++) d->xipos[3 * b + k] = pos[k] + v[k];
    real iq[4];
    quat_mul(iq, quat, MF(m, BODY_IQUAT) + 4 * b);
    quat2mat(d->ximat + 9 * b, iq);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = MI(m, SITE_BODYID)[s];
    real v[3];
    mat_vec(v, d->xmat + 9 * b, MF(m, SITE_POS) + 3 * s);
    for (int k = 0; k < 3; k++) d->site_xpos[3 * s + k] = d->xpos[3 * b + k] + v[k];
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = MI(m, GEOM_BODYID)[g];
    real v[3], q[4];
    mat_vec(v, d->xmat + 9 * b, MF(m, GEOM_POS) + 3 * g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3 * g + k] = d->xpos[3 * b + k] + v[k];
    quat_mul(q, d->xquat + 4 * b, MF(m, GEOM_QUAT) + 4 * g);
    quat2mat(d->geom_xmat + 9 * g, q);
  }
}

/* subtree COM, body inertias about the tree's COM (world axes), dof motion axes */
static void mmo_com_pos(const mmo_model* m, mmo_data* d) {
  int nb = m->nbody;
  const int mk = smark(d);
  real* sm = salloc(d, nb);
  for (int b = 0; b < nb; b++) {
    sm[b] = b == d->bmass_id ? d->bmass_val : MF(m, BODY_MASS)[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = sm[b] * d->xipos[3 * b + k];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = MI(m, BODY_PARENT)[b];
    sm[p] += sm[b];
    for (int k = 0; k < 3; k++) d->subtree_com[3 * p + k] += d->subtree_com[3 * b + k];
  }
  for (int b = 0; b < nb; b++) {
    if (sm[b] < MINVAL) for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] = d->xipos[3 * b + k];
    else for (int k = 0; k < 3; k++) d->subtree_com[3 * b + k] /= sm[b];
  }
  srelease(d, mk);
  for (int b = 1; b < nb; b++) {
    const real* c = d->subtree_com + 3 * MI(m, BODY_ROOTID)[b];
    const real* R = d->ximat + 9 * b;
    const real* I = MF(m, BODY_INERTIA) + 3 * b;
    real ms = b == d->bmass_id ? d->bmass_val : MF(m, BODY_MASS)[b];
    real r[3] = {d->xipos[3 * b] - c[0], d->xipos[3 * b + 1] - c[1], d->xipos[3 * b + 2] - c[2]};
    real* ci = d->cinert + 10 * b;
    /* R diag(I) R^T */
    real xx = 0, yy = 0, zz = 0, xy = 0, xz = 0, yz = 0;
    for (int k = 0; k < 3; k++) {
      xx += R[k] * I[k] * R[k]; yy += R[3 + k] * I[k] * R[3 + k]; zz += R[6 + k] * I[k] * R[6 + k];
      xy += R[k] * I[k] * R[3 + k]; xz += R[k] * I[k] * R[6 + k]; yz += R[3 + k] * I[k] * R[6 + k];
    }
    real r2 = dot3(r, r);
    ci[0] = xx + ms * (r2 - r[0] * r[0]); ci[1] = yy + ms * (r2 - r[1] * r[1]); ci[2] = zz + ms * (r2 - r[2] * r[2]);
    ci[3] = xy - ms * r[0] * r[1]; ci[4] = xz - ms * r[0] * r[2]; ci[5] = yz - ms * r[1] * r[2];
    ci[6] = ms * r[0]; ci[7] = ms * r[1]; ci[8] = ms * r[2]; ci[9] = ms;
  }
  memset(d->cinert, 0, 10 * sizeof(real));
  for (int j = 0; j < m->njnt; j++) {
    int b = MI(m, JNT_BODYID)[j], da = MI(m, JNT_DOFADR)[j], type = MI(m, JNT_TYPE)[j];
    const real* c = d->subtree_com + 3 * MI(m, BODY_ROOTID)[b];
    real off[3];
    for (int k = 0; k < 3; k++) off[k] = c[k] - d->xanchor[3 * j + k];
    if (type == MM_JNT_HINGE) {
      real* cd = d->cdof + 6 * da;
      for (int k = 0; k < 3; k++) cd[k] = d->xaxis[3 * j + k];
      cross3(cd + 3, cd, off);
    } else if (type == MM_JNT_SLIDE) {
      real* cd = d->cdof + 6 * da;
      cd[0] = cd[1] = cd[2] = 0;
      for (int k = 0; k < 3; k++) cd[3 + k] = d->xaxis[3 * j + k];
    } else {
      int r0 = da;
      if (type == MM_JNT_FREE) {
        for (int a = 0; a < 3; a++) {
          real* cd = d->cdof + 6 * (da + a);
          memset(cd, 0, 6 * sizeof(real));
          cd[3 + a] = 1;
        }
        r0 = da + 3;
      }
      const real* R = d->xmat + 9 * b;
      for (int a = 0; a < 3; a++) {
        real* cd = d->cdof + 6 * (r0 + a);
        cd[0] = R[a]; cd[1] = R[3 + a]; cd[2] = R[6 + a];
        cross3(cd + 3, cd, off);
      }
    }
  }
}

/* translational Jacobian (3 x nv, row-major) of world point `pnt` fixed to `body` */
static void mmo_jacp(const mmo_model* m, const mmo_data* d, real* jacp, const real* pnt, int body) {
  int nv = m->nv;
  memset(jacp, 0, sizeof(real) * 3 * nv);
  if (body <= 0) return;
  const real* c = d->subtree_com + 3 * MI(m, BODY_ROOTID)[body];
  real off[3] = {pnt[0] - c[0], pnt[1] - c[1], pnt[2] - c[2]};
  int b = body;
  while (b > 0) {
    int da = MI(m, BODY_DOFADR)[b], dn = MI(m, BODY_DOFNUM)[b];
    for (int i = da; i < da + dn; i++) {
      const real* cd = d->cdof + 6 * i;
      real t[3];
      cross3(t, cd, off);
      for (int k = 0; k < 3; k++) jacp[k * nv + i] = cd[3 + k] + t[k];
    }
    b = MI(m, BODY_PARENT)[b];
  }
}

/* full 6-D Jacobian rows: jacp (3 x nv) and jacr (3 x nv) */
static void mmo_jac(const mmo_model* m, const mmo_data* d, real* jacp, real* jacr, const real* pnt, int body) {
  int nv = m->nv;
  mmo_jacp(m, d, jacp, pnt, body);
  memset(jacr, 0, sizeof(real) * 3 * nv);
  int b = body;
  while (b > 0) {
    int da = MI(m, BODY_DOFADR)[b], dn = MI(m, BODY_DOFNUM)[b];
    for (int i = da; i < da + dn; i++)
      for (int k = 0; k < 3; k++) jacr[k * nv + i] = d->cdof[6 * i + k];
    b = MI(m, BODY_PARENT)[b];
  }
}

/* ------------------------------------------------- A2 tendon wrapping */
static int seg_intersect(const real* p1, const real* p2, const real* p3, const real* p4) {
  real det = (p4[1] - p3[1]) * (p2[0] - p1[0]) - (p4[0] - p3[0]) * (p2[1] - p1[1]);
  /* (nearly) parallel segments never cross: relative test, identical to the HIP engine's */
  real n12 = (p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1]);
  real n34 = (p4[0] - p3[0]) * (p4[0] - p3[0]) + (p4[1] - p3[1]) * (p4[1] - p3[1]);
  if (fabs(det) < MINVAL || det * det < 4e-6 * n12 * n34) return 0;
  real a = ((p4[0] - p3[0]) * (p1[1] - p3[1]) - (p4[1] - p3[1]) * (p1[0] - p3[0])) / det;
  real b = ((p2[0] - p1[0]) * (p1[1] - p3[1]) - (p2[1] - p1[1]) * (p1[0] - p3[0])) / det;
  return (a >= 0 && a <= 1 && b >= 0 && b <= 1);
}

/* 2-D: wrap segment d0->d1 around circle of `radius` at the origin.  sd: unit side
   direction or NULL.  Returns arc length, or -1 for no wrap; pnt = two tangent points. */
static real wrap_circle(real pnt[4], const real d0[2], const real d1[2], const real* sd, real radius) {
  real sqlen0 = d0[0] * d0[0] + d0[1] * d0[1], sqlen1 = d1[0] * d1[0] + d1[1] * d1[1];
  real sqrad = radius * radius;
  real dif[2] = {d1[0] - d0[0], d1[1] - d0[1]};
  real dd = dif[0] * dif[0] + dif[1] * dif[1];
  real a = -(dif[0] * d0[0] + dif[1] * d0[1]) / (dd > MINVAL ? dd : MINVAL);
  a = a < 0 ? 0 : (a > 1 ? 1 : a);
  real tmp[2] = {d0[0] + a * dif[0], d0[1] + a * dif[1]};
  if (tmp[0] * tmp[0] + tmp[1] * tmp[1] > sqrad && (!sd || sd[0] * tmp[0] + sd[1] * tmp[1] >= 0)) return -1;
  if (sqlen0 < sqrad || sqlen1 < sqrad) return -1;
  real sqrt0 = sqrt(sqlen0 - sqrad), sqrt1 = sqrt(sqlen1 - sqrad);
  real sol[2][4], good[2];
  for (int i = 0; i < 2; i++) {
    real sgn = i == 0 ? 1.0 : -1.0;
    sol[i][0] = (d0[0] * sqrad + sgn * radius * d0[1] * sqrt0) / sqlen0;
    sol[i][1] = (d0[1] * sqrad - sgn * radius * d0[0] * sqrt0) / sqlen0;
    sol[i][2] = (d1[0] * sqrad - sgn * radius * d1[1] * sqrt1) / sqlen1;
    sol[i][3] = (d1[1] * sqrad + sgn * radius * d1[0] * sqrt1) / sqlen1;
    if (sd) {
      real t[2] = {sol[i][0] + sol[i][2], sol[i][1] + sol[i][3]};
      real n = sqrt(t[0] * t[0] + t[1] * t[1]);
      if (n < MINVAL) n = MINVAL;
      good[i] = (t[0] * sd[0] + t[1] * sd[1]) / n;
    } else {
      real t[2] = {sol[i][0] - sol[i][2], sol[i][1] - sol[i][3]};
      good[i] = -(t[0] * t[0] + t[1] * t[1]);
    }
    if (seg_intersect(d0, sol[i], d1, sol[i] + 2)) good[i] = -10000;
  }
  int i = good[0] > good[1] ? 0 : 1;
  memcpy(pnt, sol[i], 4 * sizeof(real));
  if (seg_intersect(d0, pnt, d1, pnt + 2)) return -1;
  real c = (pnt[0] * pnt[2] + pnt[1] * pnt[3]) / sqrad;
  c = c > 1 ? 1 : (c < -1 ? -1 : c);
  return radius * acos(c);
}

/* 3-D wrap over sphere / cylinder (cylinder axis = local z).  Returns wrapped arc
   length (>=0) and the two surface points wpnt[6] in world coords, or -1. */
static real mmo_wrap(real wpnt[6], const real* x0, const real* x1, const real* gpos, const real* gmat,
                     real radius, int is_cyl, const real* side) {
  real t[3], p0[3], p1[3];
  for (int k = 0; k < 3; k++) t[k] = x0[k] - gpos[k];
  matT_vec(p0, gmat, t);
  for (int k = 0; k < 3; k++) t[k] = x1[k] - gpos[k];
  matT_vec(p1, gmat, t);
  if (norm3(p0) < MINVAL || norm3(p1) < MINVAL) return -1;
  real ax0[3], ax1[3];
  if (is_cyl) {
    ax0[0] = 1; ax0[1] = 0; ax0[2] = 0; ax1[0] = 0; ax1[1] = 1; ax1[2] = 0;
  } else {
    real nrm[3];
    memcpy(ax0, p0, sizeof(ax0));
    normalize3(ax0);
    cross3(nrm, p0, p1);
    if (norm3(nrm) < MINVAL) {
      /* parallel: any direction orthogonal to ax0 */
      int k = 0;
      if (fabs(ax0[1]) < fabs(ax0[k])) k = 1;
      if (fabs(ax0[2]) < fabs(ax0[k])) k = 2;
      real e[3] = {0, 0, 0};
      e[k] = 1;
      cross3(nrm, ax0, e);
    }
    normalize3(nrm);
    cross3(ax1, nrm, ax0);
    normalize3(ax1);
  }
  real d0[2] = {dot3(p0, ax0), dot3(p0, ax1)}, d1[2] = {dot3(p1, ax0), dot3(p1, ax1)};
  real sd[2], *sdp = NULL;
  if (side) {
    real s[3];
    for (int k = 0; k < 3; k++) t[k] = side[k] - gpos[k];
    matT_vec(s, gmat, t);
    sd[0] = dot3(s, ax0); sd[1] = dot3(s, ax1);
    real n = sqrt(sd[0] * sd[0] + sd[1] * sd[1]);
    if (n < MINVAL) n = MINVAL;
    sd[0] /= n; sd[1] /= n;
    sdp = sd;
  }
  real pnt[4];
  real wlen = wrap_circle(pnt, d0, d1, sdp, radius);
  if (wlen < 0) return -1;
  real r0[3], r1[3];
  for (int k = 0; k < 3; k++) {
    r0[k] = ax0[k] * pnt[0] + ax1[k] * pnt[1];
    r1[k] = ax0[k] * pnt[2] + ax1[k] * pnt[3];
  }
  if (is_cyl) {
    real L0 = sqrt((p0[0] - pnt[0]) * (p0[0] - pnt[0]) + (p0[1] - pnt[1]) * (p0[1] - pnt[1]));
    real L1 = sqrt((p1[0] - pnt[2]) * (p1[0] - pnt[2]) + (p1[1] - pnt[3]) * (p1[1] - pnt[3]));
    real tot = L0 + wlen + L1;
    if (tot < MINVAL) tot = MINVAL;
    r0[2] = p0[2] + (p1[2] - p0[2]) * L0 / tot;
    r1[2] = p0[2] + (p1[2] - p0[2]) * (L0 + wlen) / tot;
    real h = fabs(r1[2] - r0[2]);
    wlen = sqrt(wlen * wlen + h * h);
  }
  mat_vec(wpnt, gmat, r0);
  mat_vec(wpnt + 3, gmat, r1);
  for (int k = 0; k < 3; k++) { wpnt[k] += gpos[k]; wpnt[3 + k] += gpos[k]; }
  return wlen;
}

static void mmo_tendon(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  const int mk = smark(d);
  real* j0 = salloc(d, 3 * nv * 2);
  real* j1 = j0 + 3 * nv;
  memset(d->ten_J, 0, sizeof(real) * m->ntendon * nv);
  const int32_t *wt = MI(m, WRAP_TYPE), *wo = MI(m, WRAP_OBJID);
  const real* wp = MF(m, WRAP_PRM);
  for (int t = 0; t < m->ntendon; t++) {
    int adr = MI(m, TENDON_ADR)[t], num = MI(m, TENDON_NUM)[t];
    real L = 0, divisor = 1;
    real* J = d->ten_J + t * nv;
    int j = 0;
    /* fixed tendon terms */
    for (int k = 0; k < num; k++)
      if (wt[adr + k] == MM_WRAP_JOINT) {
        int jn = wo[adr + k];
        L += wp[adr + k] * d->qpos[MI(m, JNT_QPOSADR)[jn]];
        J[MI(m, JNT_DOFADR)[jn]] += wp[adr + k];
      }
    while (j < num - 1) {
      int t0 = wt[adr + j], t1 = wt[adr + j + 1];
      if (t0 == MM_WRAP_JOINT) { j++; continue; }
      if (t0 == MM_WRAP_PULLEY || t1 == MM_WRAP_PULLEY) {
        if (t0 == MM_WRAP_PULLEY) divisor = wp[adr + j];
        j++;
        continue;
      }
      real pnt[4][3];
      int body[4], npnt;
      real wlen = -1;
      int s0 = wo[adr + j];
      memcpy(pnt[0], d->site_xpos + 3 * s0, 3 * sizeof(real));
      body[0] = MI(m, SITE_BODYID)[s0];
      if (t1 == MM_WRAP_SITE) {
        int s1 = wo[adr + j + 1];
        memcpy(pnt[1], d->site_xpos + 3 * s1, 3 * sizeof(real));
        body[1] = MI(m, SITE_BODYID)[s1];
        npnt = 2;
        j += 1;
      } else {
        int g = wo[adr + j + 1], s1 = wo[adr + j + 2];
        int sideid = (int)lround(wp[adr + j + 1]);
        real w[6];
        wlen = mmo_wrap(w, pnt[0], d->site_xpos + 3 * s1, d->geom_xpos + 3 * g, d->geom_xmat + 9 * g,
                        MF(m, GEOM_SIZE)[3 * g], t1 == MM_WRAP_CYLINDER,
                        sideid >= 0 ? d->site_xpos + 3 * sideid : NULL);
        if (wlen < 0) {
          memcpy(pnt[1], d->site_xpos + 3 * s1, 3 * sizeof(real));
          body[1] = MI(m, SITE_BODYID)[s1];
          npnt = 2;
        } else {
          memcpy(pnt[1], w, 3 * sizeof(real));
          memcpy(pnt[2], w + 3, 3 * sizeof(real));
          memcpy(pnt[3], d->site_xpos + 3 * s1, 3 * sizeof(real));
          body[1] = body[2] = MI(m, GEOM_BODYID)[g];
          body[3] = MI(m, SITE_BODYID)[s1];
          npnt = 4;
        }
        j += 2;
      }
      for (int k = 0; k < npnt - 1; k++) {
        if (npnt == 4 && k == 1) { L += wlen / divisor; continue; }
        real dif[3] = {pnt[k + 1][0] - pnt[k][0], pnt[k + 1][1] - pnt[k][1], pnt[k + 1][2] - pnt[k][2]};
        real len = norm3(dif);
        L += len / divisor;
        if (len < MINVAL) { dif[0] = 1; dif[1] = dif[2] = 0; }
        else for (int c = 0; c < 3; c++) dif[c] /= len;
        if (body[k] != body[k + 1]) {
          mmo_jacp(m, d, j0, pnt[k], body[k]);
          mmo_jacp(m, d, j1, pnt[k + 1], body[k + 1]);
          for (int i = 0; i < nv; i++) {
            real s = 0;
            for (int c = 0; c < 3; c++) s += dif[c] * (j1[c * nv + i] - j0[c * nv + i]);
            J[i] += s / divisor;
          }
        }
      }
    }
    d->ten_length[t] = L;
  }
  srelease(d, mk);
}

/* A3 */
static void mmo_transmission(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  memset(d->actuator_moment, 0, sizeof(real) * m->nu * nv);
  for (int a = 0; a < m->nu; a++) {
    real gear = MF(m, ACT_GEAR)[a];
    int id = MI(m, ACT_TRNID)[a];
    if (MI(m, ACT_TRNTYPE)[a] == MM_TRN_TENDON) {
      d->actuator_length[a] = gear * d->ten_length[id];
      for (int i = 0; i < nv; i++) d->actuator_moment[a * nv + i] = gear * d->ten_J[id * nv + i];
    } else {
      d->actuator_length[a] = gear * d->qpos[MI(m, JNT_QPOSADR)[id]];
      d->actuator_moment[a * nv + MI(m, JNT_DOFADR)[id]] = gear;
    }
  }
}

/* ------------------------------------------------------ A4 inertia */
static void mmo_crb(const mmo_model* m, mmo_data* d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(real) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = MI(m, BODY_PARENT)[b];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * b + k];
  }
  memset(d->qM, 0, sizeof(real) * m->nM);
  for (int i = 0; i < nv; i++) {
    real buf[6];
    inert_mul(buf, d->crb + 10 * MI(m, DOF_BODYID)[i], d->cdof + 6 * i);
    int adr = MI(m, DOF_MADR)[i];
    d->qM[adr] = MF(m, DOF_ARMATURE)[i];
    int j = i;
    while (j >= 0) {
      real s = 0;
      for (int k = 0; k < 6; k++) s += d->cdof[6 * j + k] * buf[k];
      d->qM[adr] += s;
      adr++;
      j = MI(m, DOF_PARENTID)[j];
    }
  }
}

/* in-place sparse L' D L factorisation over the dof tree.  LD has M's layout:
   row i = [D_i, L(i,parent), L(i,grandparent), ...]; diaginv = 1/D. */
static void mmo_factor(const mmo_model* m, real* LD, real* diaginv) {
  int nv = m->nv;
  const int32_t *par = MI(m, DOF_PARENTID), *madr = MI(m, DOF_MADR);
  for (int k = nv - 1; k >= 0; k--) {
    real Mkk = LD[madr[k]];
    if (Mkk < MINVAL) Mkk = MINVAL;
    diaginv[k] = 1.0 / Mkk;
    int i = par[k], ai = 1;
    while (i >= 0) {
      real tmp = LD[madr[k] + ai] / Mkk; /* L(k,i) */
      /* row i -= tmp * (tail of row k starting at i) */
      int j = i, aj = 0;
      while (j >= 0) {
        LD[madr[i] + aj] -= tmp * LD[madr[k] + ai + aj];
        j = par[j]; aj++;
      }
      LD[madr[k] + ai] = tmp;
      i = par[i]; ai++;
    }
  }
}

/* x <- (L' D L)^{-1} x */
static void mmo_solve_ld(const mmo_model* m, const real* LD, const real* diaginv, real* x) {
  int nv = m->nv;
  const int32_t *par = MI(m, DOF_PARENTID), *madr = MI(m, DOF_MADR);
  /* x <- inv(L') x */
  for (int i = nv - 1; i >= 0; i--) {
    int j = par[i], a = 1;
    while (j >= 0) { x[j] -= LD[madr[i] + a] * x[i]; j = par[j]; a++; }
  }
  for (int i = 0; i < nv; i++) x[i] *= diaginv[i];
  /* x <- inv(L) x */
  for (int i = 0; i < nv; i++) {
    int j = par[i], a = 1;
    while (j >= 0) { x[i] -= LD[madr[i] + a] * x[j]; j = par[j]; a++; }
  }
}

/* y = M x (sparse, symmetric) */
static void mmo_mul_m(const mmo_model* m, const real* M, real* y, const real* x) {
  int nv = m->nv;
  const int32_t *par = MI(m, DOF_PARENTID), *madr = MI(m, DOF_MADR);
  for (int i = 0; i < nv; i++) y[i] = 0;
  for (int i = 0; i < nv; i++) {
    y[i] += M[madr[i]] * x[i];
    int j = par[i], a = 1;
    while (j >= 0) {
      y[i] += M[madr[i] + a] * x[j];
      y[j] += M[madr[i] + a] * x[i];
      j = par[j]; a++;
    }
  }
}

/* -------------------------------------------------- A5 velocity stage */
static void mmo_com_vel(const mmo_model* m, mmo_data* d) {
  memset(d->cvel, 0, 6 * sizeof(real));
  for (int b = 1; b < m->nbody; b++) {
    real cvel[6];
    memcpy(cvel, d->cvel + 6 * MI(m, BODY_PARENT)[b], sizeof(cvel));
    int ja = MI(m, BODY_JNTADR)[b], jn = MI(m, BODY_JNTNUM)[b];
    for (int j = ja; j < ja + jn; j++) {
      int type = MI(m, JNT_TYPE)[j], da = MI(m, JNT_DOFADR)[j];
      if (type == MM_JNT_FREE) {
        memset(d->cdof_dot + 6 * da, 0, 18 * sizeof(real));
        for (int a = 0; a < 3; a++)
          for (int k = 0; k < 6; k++) cvel[k] += d->cdof[6 * (da + a) + k] * d->qvel[da + a];
        da += 3;
        type = MM_JNT_BALL;
      }
      if (type == MM_JNT_BALL) {
        for (int a = 0; a < 3; a++) cross_motion(d->cdof_dot + 6 * (da + a), cvel, d->cdof + 6 * (da + a));
        for (int a = 0; a < 3; a++)
          for (int k = 0; k < 6; k++) cvel[k] += d->cdof[6 * (da + a) + k] * d->qvel[da + a];
      } else {
        cross_motion(d->cdof_dot + 6 * da, cvel, d->cdof + 6 * da);
        for (int k = 0; k < 6; k++) cvel[k] += d->cdof[6 * da + k] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof(cvel));
  }
}

static void mmo_passive(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -MF(m, DOF_DAMPING)[i] * d->qvel[i];
  for (int j = 0; j < m->njnt; j++) {
    real k = MF(m, JNT_STIFFNESS)[j];
    int type = MI(m, JNT_TYPE)[j];
    if (k == 0 || (type != MM_JNT_HINGE && type != MM_JNT_SLIDE)) continue;
    int qa = MI(m, JNT_QPOSADR)[j];
    d->qfrc_passive[MI(m, JNT_DOFADR)[j]] -= k * (d->qpos[qa] - MF(m, QPOS_SPRING)[qa]);
  }
  for (int t = 0; t < m->ntendon; t++) {
    real k = MF(m, TENDON_STIFFNESS)[t], b = MF(m, TENDON_DAMPING)[t];
    if (k == 0 && b == 0) continue;
    real L = d->ten_length[t], lo = MF(m, TENDON_LENGTHSPRING)[2 * t], hi = MF(m, TENDON_LENGTHSPRING)[2 * t + 1];
    real f = 0;
    if (L > hi) f = k * (hi - L);
    else if (L < lo) f = k * (lo - L);
    f -= b * d->ten_velocity[t];
    for (int i = 0; i < nv; i++) d->qfrc_passive[i] += d->ten_J[t * nv + i] * f;
  }
}

/* bias force: RNE with zero joint acceleration */
static void mmo_rne(const mmo_model* m, mmo_data* d) {
  int nb = m->nbody;
  real* cacc = d->cacc; real* cfrc = d->cfrc;
  cacc[0] = cacc[1] = cacc[2] = 0;
  for (int k = 0; k < 3; k++) cacc[3 + k] = -m->gravity[k];
  memset(cfrc, 0, 6 * sizeof(real));
  for (int b = 1; b < nb; b++) {
    int da = MI(m, BODY_DOFADR)[b], dn = MI(m, BODY_DOFNUM)[b];
    real* a = cacc + 6 * b;
    memcpy(a, cacc + 6 * MI(m, BODY_PARENT)[b], 6 * sizeof(real));
    for (int i = da; i < da + dn; i++)
      for (int k = 0; k < 6; k++) a[k] += d->cdof_dot[6 * i + k] * d->qvel[i];
    real Ia[6], Iv[6], x[6];
    inert_mul(Ia, d->cinert + 10 * b, a);
    inert_mul(Iv, d->cinert + 10 * b, d->cvel + 6 * b);
    cross_force(x, d->cvel + 6 * b, Iv);
    for (int k = 0; k < 6; k++) cfrc[6 * b + k] = Ia[k] + x[k];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = MI(m, BODY_PARENT)[b];
    if (p > 0) for (int k = 0; k < 6; k++) cfrc[6 * p + k] += cfrc[6 * b + k];
  }
  for (int i = 0; i < m->nv; i++) {
    real s = 0;
    const real* f = cfrc + 6 * MI(m, DOF_BODYID)[i];
    for (int k = 0; k < 6; k++) s += d->cdof[6 * i + k] * f[k];
    d->qfrc_bias[i] = s;
  }
}

/* ------------------------------------------------- A6 muscle actuation */
static real muscle_fl(real L, real lmin, real lmax) {
  if (L < lmin || L > lmax) return 0;
  real a = 0.5 * (lmin + 1), b = 0.5 * (1 + lmax), x;
  if (L <= a) { x = (L - lmin) / fmax(MINVAL, a - lmin); return 0.5 * x * x; }
  if (L <= 1) { x = (1 - L) / fmax(MINVAL, 1 - a); return 1 - 0.5 * x * x; }
  if (L <= b) { x = (L - 1) / fmax(MINVAL, b - 1); return 1 - 0.5 * x * x; }
  x = (lmax - L) / fmax(MINVAL, lmax - b);
  return 0.5 * x * x;
}
static real muscle_f0(const real* prm, real acc0) {
  return prm[2] >= 0 ? prm[2] : prm[3] / fmax(MINVAL, acc0);
}
static real muscle_gain(real len, real vel, const real* lr, real acc0, const real* prm) {
  real force = muscle_f0(prm, acc0);
  real L0 = (lr[1] - lr[0]) / fmax(MINVAL, prm[1] - prm[0]);
  real L = prm[0] + (len - lr[0]) / fmax(MINVAL, L0);
  real V = vel / fmax(MINVAL, L0 * prm[6]);
  real FL = muscle_fl(L, prm[4], prm[5]);
  real fvmax = prm[8], y = fvmax - 1, FV;
  if (V <= -1) FV = 0;
  else if (V <= 0) FV = (V + 1) * (V + 1);
  else if (V <= y) FV = fvmax - (y - V) * (y - V) / fmax(MINVAL, y);
  else FV = fvmax;
  return -force * FL * FV;
}
static real muscle_bias(real len, const real* lr, real acc0, const real* prm) {
  real force = muscle_f0(prm, acc0);
  real L0 = (lr[1] - lr[0]) / fmax(MINVAL, prm[1] - prm[0]);
  real L = prm[0] + (len - lr[0]) / fmax(MINVAL, L0);
  real b = 0.5 * (1 + prm[5]), fpmax = prm[7], x;
  if (L <= 1) return 0;
  if (L <= b) { x = (L - 1) / fmax(MINVAL, b - 1); return -force * fpmax * 0.5 * x * x; }
  x = (L - b) / fmax(MINVAL, b - 1);
  return -force * fpmax * (0.5 + x);
}
static real sigmoid5(real x) {
  if (x <= 0) return 0;
  if (x >= 1) return 1;
  return x * x * x * (3 * x * (2 * x - 5) + 10);
}
static real muscle_dynamics(real ctrl, real act, const real* prm) {
  real cc = ctrl < 0 ? 0 : (ctrl > 1 ? 1 : ctrl);
  real ac = act < 0 ? 0 : (act > 1 ? 1 : act);
  real tau_act = prm[0] * (0.5 + 1.5 * ac), tau_deact = prm[1] / (0.5 + 1.5 * ac);
  real dctrl = cc - act, tau;
  if (prm[2] < MINVAL) tau = dctrl > 0 ? tau_act : tau_deact;
  else tau = tau_deact + (tau_act - tau_deact) * sigmoid5(dctrl / prm[2] + 0.5);
  return dctrl / fmax(MINVAL, tau);
}

static void mmo_actuation(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  memset(d->qfrc_actuator, 0, sizeof(real) * nv);
  /* mj_fwdActuation: "check controls, set all to 0 if any are bad" (mjWARN_BADCTRL; bad = NaN, or beyond mjMAXVAL = 1e10) */
  for (int a = 0; a < m->nu; a++)
    if (!(fabs(d->ctrl[a]) < 1e10)) { memset(d->ctrl, 0, sizeof(real) * m->nu); d->warn_bad |= 32; break; }   /* (32 = the kernel's status bit; 4 is the nconmax overflow) */
  for (int a = 0; a < m->nu; a++) {
    real ctrl = d->ctrl[a];
    if (MI(m, ACT_CTRLLIMITED)[a]) {
      real lo = MF(m, ACT_CTRLRANGE)[2 * a], hi = MF(m, ACT_CTRLRANGE)[2 * a + 1];
      ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
    }
    int aa = MI(m, ACT_ACTADR)[a];
    real input = ctrl;
    if (MI(m, ACT_DYNTYPE)[a] == MM_DYN_MUSCLE) {
      d->act_dot[aa] = muscle_dynamics(ctrl, d->act[aa], MF(m, ACT_DYNPRM) + 3 * a);
      input = d->act[aa];
    } else if (MI(m, ACT_DYNTYPE)[a] == MM_DYN_INTEGRATOR) {
      d->act_dot[aa] = ctrl; input = d->act[aa];
    } else if (MI(m, ACT_DYNTYPE)[a] == MM_DYN_FILTER) {
      d->act_dot[aa] = (ctrl - d->act[aa]) / fmax(MINVAL, MF(m, ACT_DYNPRM)[3 * a]); input = d->act[aa];
    }
    real gain, bias = 0;
    const real* lr = MF(m, ACT_LENGTHRANGE) + 2 * a;
    if (MI(m, ACT_GAINTYPE)[a] == MM_GAIN_MUSCLE)
      gain = muscle_gain(d->actuator_length[a], d->actuator_velocity[a], lr, MF(m, ACT_ACC0)[a],
                         MF(m, ACT_GAINPRM) + 9 * a);
    else gain = MF(m, ACT_GAINPRM)[9 * a];
    if (MI(m, ACT_BIASTYPE)[a] == MM_BIAS_MUSCLE)
      bias = muscle_bias(d->actuator_length[a], lr, MF(m, ACT_ACC0)[a], MF(m, ACT_BIASPRM) + 9 * a);
    else if (MI(m, ACT_BIASTYPE)[a] == MM_BIAS_AFFINE) {   /* position / velocity servos: mjBIAS_AFFINE */
      const real* bp = MF(m, ACT_BIASPRM) + 9 * a;
      bias = bp[0] + bp[1] * d->actuator_length[a] + bp[2] * d->actuator_velocity[a];
    }
    real f = gain * input + bias;
    if (MI(m, ACT_FORCELIMITED)[a]) {
      real lo = MF(m, ACT_FORCERANGE)[2 * a], hi = MF(m, ACT_FORCERANGE)[2 * a + 1];
      f = f < lo ? lo : (f > hi ? hi : f);
    }
    d->actuator_force[a] = f;
    for (int i = 0; i < nv; i++) d->qfrc_actuator[i] += d->actuator_moment[a * nv + i] * f;
  }
}

/* ------------------------------------------------- A7 constraints */
static int add_row(const mmo_model* m, mmo_data* d, int type, int id, real pos, real margin, real diagApprox,
                   const real* solref, const real* solimp) {
  int r = d->nefc;
  if (r >= m->njmax) { d->warn_bad |= 2; return -1; }
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  d->efc_diagApprox[r] = diagApprox; d->efc_floss[r] = 0;
  memcpy(d->efc_solref + 2 * r, solref, 2 * sizeof(real));
  memcpy(d->efc_solimp + 5 * r, solimp, 5 * sizeof(real));
  memset(d->efc_J + r * m->nv, 0, sizeof(real) * m->nv);
  d->nefc++;
  return r;
}

#include "mmo_collision.inc"

static void mmo_make_constraint(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  d->nefc = 0; d->ncon = 0;
  /* equality: joint coupling  q1 - q1_0 = poly(q2 - q2_0) */
  for (int e = 0; e < m->neq; e++) {
    int j1 = MI(m, EQ_OBJ1ID)[e], j2 = MI(m, EQ_OBJ2ID)[e];
    const real* c = MF(m, EQ_DATA) + 5 * e;
    int q1 = MI(m, JNT_QPOSADR)[j1], d1 = MI(m, JNT_DOFADR)[j1];
    real pos1 = d->qpos[q1] - MF(m, QPOS0)[q1];
    real res, deriv = 0;
    if (j2 >= 0) {
      int q2 = MI(m, JNT_QPOSADR)[j2];
      real x = d->qpos[q2] - MF(m, QPOS0)[q2];
      real poly = c[0] + x * (c[1] + x * (c[2] + x * (c[3] + x * c[4])));
      deriv = c[1] + x * (2 * c[2] + x * (3 * c[3] + x * 4 * c[4]));
      res = pos1 - poly;
    } else res = pos1 - c[0];
    real dA = MF(m, DOF_INVWEIGHT0)[d1];
    if (j2 >= 0) dA += MF(m, DOF_INVWEIGHT0)[MI(m, JNT_DOFADR)[j2]];
    int r = add_row(m, d, MM_CON_EQUALITY, e, res, 0, dA, MF(m, EQ_SOLREF) + 2 * e, MF(m, EQ_SOLIMP) + 5 * e);
    if (r < 0) continue;
    d->efc_J[r * nv + d1] = 1;
    if (j2 >= 0) d->efc_J[r * nv + MI(m, JNT_DOFADR)[j2]] = -deriv;
  }
  /* dof friction loss (MuJoCo order: equality, friction loss, limits, contacts): a row J = e_dof at pos 0 whose force is
     bounded by +-frictionloss (Huber cost, see row_cost) */
  for (int i = 0; i < nv; i++) {
    real f = MF(m, DOF_FRICTIONLOSS)[i];
    if (!(f > 0)) continue;
    int r = add_row(m, d, MM_CON_FRICTION_DOF, i, 0, 0, MF(m, DOF_INVWEIGHT0)[i], MF(m, DOF_SOLREF) + 2 * i,
                    MF(m, DOF_SOLIMP) + 5 * i);
    if (r < 0) continue;
    d->efc_J[r * nv + i] = 1; d->efc_floss[r] = f;
  }
  /* joint limits (hinge / slide) */
  for (int j = 0; j < m->njnt; j++) {
    int type = MI(m, JNT_TYPE)[j];
    if (!MI(m, JNT_LIMITED)[j] || (type != MM_JNT_HINGE && type != MM_JNT_SLIDE)) continue;
    real q = d->qpos[MI(m, JNT_QPOSADR)[j]], margin = MF(m, JNT_MARGIN)[j];
    int dof = MI(m, JNT_DOFADR)[j];
    for (int side = -1; side <= 1; side += 2) {
      real dist = side < 0 ? q - MF(m, JNT_RANGE)[2 * j] : MF(m, JNT_RANGE)[2 * j + 1] - q;
      if (dist < margin) {
        int r = add_row(m, d, MM_CON_LIMIT_JOINT, j, dist, margin, MF(m, DOF_INVWEIGHT0)[dof],
                        MF(m, JNT_SOLREF) + 2 * j, MF(m, JNT_SOLIMP) + 5 * j);
        if (r >= 0) d->efc_J[r * nv + dof] = -(real)side;
      }
    }
  }
  /* tendon limits */
  for (int t = 0; t < m->ntendon; t++) {
    if (!MI(m, TENDON_LIMITED)[t]) continue;
    real L = d->ten_length[t], margin = MF(m, TENDON_MARGIN)[t];
    for (int side = -1; side <= 1; side += 2) {
      real dist = side < 0 ? L - MF(m, TENDON_RANGE)[2 * t] : MF(m, TENDON_RANGE)[2 * t + 1] - L;
      if (dist < margin) {
        int r = add_row(m, d, MM_CON_LIMIT_TENDON, t, dist, margin, MF(m, TENDON_INVWEIGHT0)[t],
                        MF(m, TENDON_SOLREF) + 2 * t, MF(m, TENDON_SOLIMP) + 5 * t);
        if (r >= 0) for (int i = 0; i < nv; i++) d->efc_J[r * nv + i] = -(real)side * d->ten_J[t * nv + i];
      }
    }
  }
  /* contacts (pyramidal cones) */
  mmo_collide_and_add(m, d);
}

/* impedance, regularisation, reference acceleration (needs efc_vel) */
static void mmo_reference_constraint(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  for (int r = 0; r < d->nefc; r++) {
    const real* si = d->efc_solimp + 5 * r; const real* sr = d->efc_solref + 2 * r;
    real dmin = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
    /* sanitise like MuJoCo's getsolparam */
    dmin = fmin(fmax(dmin, 0.0001), 0.9999); dmax = fmin(fmax(dmax, 0.0001), 0.9999);
    width = fmax(0, width); mid = fmin(fmax(mid, 0.0001), 0.9999); power = fmax(1, power);
    real x = d->efc_pos[r] - d->efc_margin[r], imp;
    if (width < MINVAL || dmin == dmax) imp = 0.5 * (dmin + dmax);
    else {
      real xa = fabs(x) / width, y;
      if (xa >= 1) imp = dmax;
      else if (xa == 0) imp = dmin;
      else {
        if (power == 1) y = xa;
        else if (xa <= mid) y = pow(xa, power) / pow(mid, power - 1);
        else y = 1 - pow(1 - xa, power) / pow(1 - mid, power - 1);
        imp = dmin + y * (dmax - dmin);
      }
    }
    real R = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[r] / imp);
    real K, B;
    if (sr[0] > 0) {
      real tc = fmax(sr[0], 2 * m->timestep), dr = sr[1]; /* refsafe */
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { K = -sr[0] / fmax(MINVAL, dmax * dmax); B = -sr[1] / fmax(MINVAL, dmax); }
    real vel = 0;
    for (int i = 0; i < nv; i++) vel += d->efc_J[r * nv + i] * d->qvel[i];
    d->efc_R[r] = R; d->efc_D[r] = 1 / R; d->efc_vel[r] = vel;
    d->efc_aref[r] = -B * vel - K * imp * x;
  }
  mmo_contact_pyramid_adjust(m, d);
}

/* ---- Newton solver on  1/2 (a-a0)'M(a-a0) + sum_i s_i(J_i a - aref_i) ------- */
typedef struct { real cost, d1, d2; } lspoint;

/* cost s_i(x) of row i at x = J_i a - aref_i, its force -s' and curvature s'' (MuJoCo constraint model: equality rows are
   quadratic everywhere, limit / contact rows only for x < 0, friction-loss rows are Huber: quadratic for |x| < R f, linear
   with slope -+f outside, so that the force saturates at +-f) */
static inline real row_cost(const mmo_data* d, int i, real x, real* force, real* curv) {
  real D = d->efc_D[i];
  if (d->efc_type[i] == MM_CON_FRICTION_DOF) {
    real f = d->efc_floss[i], rf = f / D;
    if (x <= -rf) { *force = f; *curv = 0; return f * (-0.5 * rf - x); }
    if (x >= rf) { *force = -f; *curv = 0; return f * (-0.5 * rf + x); }
    *force = -D * x; *curv = D; return 0.5 * D * x * x;
  }
  if (d->efc_type[i] == MM_CON_EQUALITY || x < 0) { *force = -D * x; *curv = D; return 0.5 * D * x * x; }
  *force = 0; *curv = 0; return 0;
}

static lspoint ls_eval(const mmo_data* d, int nefc, const real* jar, const real* jv, const real* quadg, real alpha) {
  lspoint p;
  p.cost = quadg[0] + alpha * (quadg[1] + alpha * quadg[2]);
  p.d1 = quadg[1] + 2 * alpha * quadg[2];
  p.d2 = 2 * quadg[2];
  for (int i = 0; i < nefc; i++) {
    real x = jar[i] + alpha * jv[i], f, c;
    p.cost += row_cost(d, i, x, &f, &c);
    p.d1 -= f * jv[i]; p.d2 += c * jv[i] * jv[i];
  }
  return p;
}

static void mmo_solve(const mmo_model* m, mmo_data* d) {
  int nv = m->nv, nefc = d->nefc;
  d->solver_niter = 0;
  if (nefc == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(real) * nv);
    memset(d->qfrc_constraint, 0, sizeof(real) * nv);
    return;
  }
  const int mk = smark(d);
  real* Ma = salloc(d, nv); real* grad = salloc(d, nv); real* Mgrad = salloc(d, nv); real* search = salloc(d, nv);
  real* Mv = salloc(d, nv); real* jar = salloc(d, nefc); real* jv = salloc(d, nefc);
  real* Hd = salloc(d, nv * nv);
  int* active = (int*)(void*)salloc(d, nefc);   /* sizeof(real) >= sizeof(int) */
  real scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));

#define EVAL_COST(qa, out)                                                            \
  do {                                                                                \
    mmo_mul_m(m, d->qM, Ma, qa);                                                      \
    real g_ = 0;                                                                      \
    for (int i_ = 0; i_ < nv; i_++) g_ += (qa[i_] - d->qacc_smooth[i_]) * (Ma[i_] - d->qfrc_smooth[i_]); \
    real c_ = 0.5 * g_;                                                               \
    for (int r_ = 0; r_ < nefc; r_++) {                                               \
      real x_ = -d->efc_aref[r_];                                                     \
      for (int i_ = 0; i_ < nv; i_++) x_ += d->efc_J[r_ * nv + i_] * qa[i_];          \
      jar[r_] = x_;                                                                   \
      { real f_, h_; c_ += row_cost(d, r_, x_, &f_, &h_); }                           \
    }                                                                                 \
    out = c_;                                                                         \
  } while (0)

  /* warm start: keep qacc_warmstart only if it beats the unconstrained solution */
  real cost_ws, cost_sm;
  EVAL_COST(d->qacc_warmstart, cost_ws);
  EVAL_COST(d->qacc_smooth, cost_sm);
  if (cost_ws < cost_sm) memcpy(d->qacc, d->qacc_warmstart, sizeof(real) * nv);
  else memcpy(d->qacc, d->qacc_smooth, sizeof(real) * nv);
  real cost;
  EVAL_COST(d->qacc, cost);

  for (int iter = 0; iter < m->iterations; iter++) {
    /* active set, forces, gradient */
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
    for (int r = 0; r < nefc; r++) {
      real curv;
      (void)row_cost(d, r, jar[r], &d->efc_force[r], &curv);
      active[r] = curv > 0;      /* quadratic state: contributes D J'J to the Hessian */
      if (d->efc_force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * d->efc_force[r];
    }
    real gnorm = 0;
    for (int i = 0; i < nv; i++) {
      grad[i] = Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
      gnorm += grad[i] * grad[i];
    }
    gnorm = sqrt(gnorm);
    if (scale * gnorm < m->tolerance) break;
    /* Hessian H = M + J_A' D J_A (dense), Cholesky solve */
    memset(Hd, 0, sizeof(real) * nv * nv);
    {
      const int32_t *par = MI(m, DOF_PARENTID), *madr = MI(m, DOF_MADR);
      for (int i = 0; i < nv; i++) {
        int j = i, a = 0;
        while (j >= 0) { Hd[i * nv + j] = Hd[j * nv + i] = d->qM[madr[i] + a]; j = par[j]; a++; }
      }
    }
    for (int r = 0; r < nefc; r++) if (active[r]) {
      const real* J = d->efc_J + r * nv; real D = d->efc_D[r];
      for (int i = 0; i < nv; i++) if (J[i] != 0)
        for (int j = 0; j < nv; j++) Hd[i * nv + j] += D * J[i] * J[j];
    }
    /* in-place Cholesky (lower) */
    for (int i = 0; i < nv; i++) {
      for (int j = 0; j <= i; j++) {
        real s = Hd[i * nv + j];
        for (int k = 0; k < j; k++) s -= Hd[i * nv + k] * Hd[j * nv + k];
        if (i == j) Hd[i * nv + i] = sqrt(s > MINVAL ? s : MINVAL);
        else Hd[i * nv + j] = s / Hd[j * nv + j];
      }
    }
    for (int i = 0; i < nv; i++) {
      real s = grad[i];
      for (int k = 0; k < i; k++) s -= Hd[i * nv + k] * Mgrad[k];
      Mgrad[i] = s / Hd[i * nv + i];
    }
    for (int i = nv - 1; i >= 0; i--) {
      real s = Mgrad[i];
      for (int k = i + 1; k < nv; k++) s -= Hd[k * nv + i] * Mgrad[k];
      Mgrad[i] = s / Hd[i * nv + i];
    }
    for (int i = 0; i < nv; i++) search[i] = -Mgrad[i];

    /* ---- exact line search on the convex piecewise-quadratic phi(alpha) ---- */
    mmo_mul_m(m, d->qM, Mv, search);
    real snorm = 0;
    for (int i = 0; i < nv; i++) snorm += search[i] * search[i];
    snorm = sqrt(snorm);
    if (snorm < MINVAL) break;
    for (int r = 0; r < nefc; r++) {
      real s = 0;
      for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * search[i];
      jv[r] = s;
    }
    real quadg[3] = {0, 0, 0};
    for (int i = 0; i < nv; i++) {
      quadg[0] += 0.5 * (d->qacc[i] - d->qacc_smooth[i]) * (Ma[i] - d->qfrc_smooth[i]);
      quadg[1] += search[i] * (Ma[i] - d->qfrc_smooth[i]);
      quadg[2] += 0.5 * search[i] * Mv[i];
    }
    real gtol = m->tolerance * m->ls_tolerance * snorm / scale;
    real alpha = 0, lo = 0, hi = -1; /* hi<0: no upper bracket yet */
    lspoint p = ls_eval(d, nefc, jar, jv, quadg, 0);
    real best_alpha = 0, best_cost = p.cost;
    for (int it = 0; it < m->ls_iterations; it++) {
      if (fabs(p.d1) < gtol) break;
      if (p.d1 < 0) lo = alpha; else hi = alpha;
      real next = alpha - p.d1 / fmax(p.d2, MINVAL);
      if (hi >= 0 && (next <= lo || next >= hi)) next = 0.5 * (lo + hi);
      else if (hi < 0 && next <= lo) next = 2 * lo + 1e-10;
      alpha = next;
      p = ls_eval(d, nefc, jar, jv, quadg, alpha);
      if (p.cost < best_cost) { best_cost = p.cost; best_alpha = alpha; }
    }
    alpha = best_alpha;
    if (alpha == 0) break;
    /* move */
    for (int i = 0; i < nv; i++) { d->qacc[i] += alpha * search[i]; Ma[i] += alpha * Mv[i]; }
    for (int r = 0; r < nefc; r++) jar[r] += alpha * jv[r];
    real old = cost;
    cost = best_cost;
    d->solver_niter = iter + 1;
    if (scale * (old - cost) < m->tolerance) {
      /* refresh forces for the final state before exiting */
      for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
      for (int r = 0; r < nefc; r++) {
        real curv;
        (void)row_cost(d, r, jar[r], &d->efc_force[r], &curv);
        if (d->efc_force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * d->efc_force[r];
      }
      break;
    }
  }
#undef EVAL_COST
  srelease(d, mk);
}

/* ------------------------------------------------------ pipeline */
void mmo_fwd_position(const mmo_model* m, mmo_data* d) {
  MMO_STAGE(MMO_ST_KIN);
  mmo_kinematics(m, d);
  MMO_STAGE(MMO_ST_COM);
  mmo_com_pos(m, d);
  MMO_STAGE(MMO_ST_TENDON);
  mmo_tendon(m, d);
  mmo_transmission(m, d);
  MMO_STAGE(MMO_ST_CRB);
  mmo_crb(m, d);
  memcpy(d->qLD, d->qM, sizeof(real) * m->nM);
  mmo_factor(m, d->qLD, d->qLDiagInv);
  MMO_STAGE(MMO_ST_CONSTR);
  mmo_make_constraint(m, d);
  MMO_STAGE(MMO_ST_NONE);
}

void mmo_fwd_velocity(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  MMO_STAGE(MMO_ST_VEL);
  for (int t = 0; t < m->ntendon; t++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->ten_J[t * nv + i] * d->qvel[i];
    d->ten_velocity[t] = s;
  }
  for (int a = 0; a < m->nu; a++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->actuator_moment[a * nv + i] * d->qvel[i];
    d->actuator_velocity[a] = s;
  }
  mmo_com_vel(m, d);
  mmo_passive(m, d);
  mmo_rne(m, d);
  MMO_STAGE(MMO_ST_CONSTR);
  mmo_reference_constraint(m, d);
  MMO_STAGE(MMO_ST_NONE);
}

void mmo_fwd_acceleration(const mmo_model* m, mmo_data* d) {
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
    d->qacc_smooth[i] = d->qfrc_smooth[i];
  }
  mmo_solve_ld(m, d->qLD, d->qLDiagInv, d->qacc_smooth);
}

void mmo_forward(const mmo_model* m, mmo_data* d) {
  mmo_fwd_position(m, d);
  mmo_fwd_velocity(m, d);
  MMO_STAGE(MMO_ST_ACT);
  mmo_actuation(m, d);
  MMO_STAGE(MMO_ST_ACC);
  mmo_fwd_acceleration(m, d);
  MMO_STAGE(MMO_ST_SOLVE);
  mmo_solve(m, d);
  MMO_STAGE(MMO_ST_NONE);
}

static int bad_state(const mmo_model* m, const mmo_data* d, int check_acc) {
  for (int i = 0; i < m->nq; i++) if (!(fabs(d->qpos[i]) < 1e10)) return 1;
  for (int i = 0; i < m->nv; i++) if (!(fabs(d->qvel[i]) < 1e10)) return 1;
  if (check_acc) for (int i = 0; i < m->nv; i++) if (!(fabs(d->qacc[i]) < 1e10)) return 1;
  return 0;
}

void mmo_full_m(const mmo_model* m, const mmo_data* d, real* out);
/* mj_advance: activations, qvel += h qacc, qpos on the manifold, time */
static void mmo_advance(const mmo_model* m, mmo_data* d, const real* qacc) {
  int nv = m->nv;
  real h = m->timestep;
  for (int a = 0; a < m->nu; a++) {
    int aa = MI(m, ACT_ACTADR)[a];
    if (aa < 0) continue;
    real x = d->act[aa] + h * d->act_dot[aa];
    if (MI(m, ACT_DYNTYPE)[a] == MM_DYN_MUSCLE) x = x < 0 ? 0 : (x > 1 ? 1 : x);
    d->act[aa] = x;
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) {
    int type = MI(m, JNT_TYPE)[j], qa = MI(m, JNT_QPOSADR)[j], da = MI(m, JNT_DOFADR)[j];
    if (type == MM_JNT_HINGE || type == MM_JNT_SLIDE) { d->qpos[qa] += h * d->qvel[da]; continue; }
    if (type == MM_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
      qa += 3; da += 3;
    }
    /* quaternion integration: q <- q * exp(h*w/2), w in the local frame */
    real w[3] = {d->qvel[da], d->qvel[da + 1], d->qvel[da + 2]};
    real ang = h * norm3(w);
    if (ang > MINVAL) {
      real ax[3] = {w[0], w[1], w[2]}, dq[4], qn[4];
      normalize3(ax);
      axisangle2quat(dq, ax, ang);
      quat_mul(qn, d->qpos + qa, dq);
      quat_normalize(qn);
      memcpy(d->qpos + qa, qn, sizeof(qn));
    }
  }
  d->time += h;
}

/* A9: semi-implicit Euler with implicit joint damping */
static void mmo_euler(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  real h = m->timestep;
  real* qacc = d->tmp_nv;
  int damped = 0;
  for (int i = 0; i < nv; i++) if (MF(m, DOF_DAMPING)[i] > 0) damped = 1;
  if (damped && m->eulerdamp) {
    memcpy(d->qH, d->qM, sizeof(real) * m->nM);
    for (int i = 0; i < nv; i++) d->qH[MI(m, DOF_MADR)[i]] += h * MF(m, DOF_DAMPING)[i];
    mmo_factor(m, d->qH, d->qHDiagInv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    mmo_solve_ld(m, d->qH, d->qHDiagInv, qacc);
  } else memcpy(qacc, d->qacc, sizeof(real) * nv);
  mmo_advance(m, d, qacc);
}

/* d(actuator force)/d(actuator velocity): mjd_actuator_vel of MuJoCo's derivatives.c (bias_vel + gain_vel * input; a force
   sitting on its forcerange has no derivative).  Muscle: gain = -F0 FL(L) FV(V), V = vel / (L0 vmax) */
static real actuator_dforce_dvel(const mmo_model* m, const mmo_data* d, int a) {
  real bias_vel = 0, gain_vel = 0;
  if (MI(m, ACT_BIASTYPE)[a] == MM_BIAS_AFFINE) bias_vel = MF(m, ACT_BIASPRM)[9 * a + 2];
  if (MI(m, ACT_GAINTYPE)[a] == MM_GAIN_MUSCLE) {
    const real* prm = MF(m, ACT_GAINPRM) + 9 * a; const real* lr = MF(m, ACT_LENGTHRANGE) + 2 * a;
    real force = muscle_f0(prm, MF(m, ACT_ACC0)[a]);
    real L0 = (lr[1] - lr[0]) / fmax(MINVAL, prm[1] - prm[0]);
    real L = prm[0] + (d->actuator_length[a] - lr[0]) / fmax(MINVAL, L0);
    real vs = fmax(MINVAL, L0 * prm[6]), V = d->actuator_velocity[a] / vs;
    real FL = muscle_fl(L, prm[4], prm[5]);
    real fvmax = prm[8], y = fvmax - 1, dFV;
    if (V <= -1) dFV = 0;
    else if (V <= 0) dFV = 2 * (V + 1);
    else if (V <= y) dFV = 2 * (y - V) / fmax(MINVAL, y);
    else dFV = 0;
    gain_vel = -force * FL * dFV / vs;
  }
  if (MI(m, ACT_FORCELIMITED)[a]) {
    real f = d->actuator_force[a];
    if (f <= MF(m, ACT_FORCERANGE)[2 * a] || f >= MF(m, ACT_FORCERANGE)[2 * a + 1]) return 0;
  }
  int aa = MI(m, ACT_ACTADR)[a];
  real input = aa >= 0 ? d->act[aa] : d->ctrl[a];
  if (aa < 0 && MI(m, ACT_CTRLLIMITED)[a]) {
    real lo = MF(m, ACT_CTRLRANGE)[2 * a], hi = MF(m, ACT_CTRLRANGE)[2 * a + 1];
    input = input < lo ? lo : (input > hi ? hi : input);
  }
  return bias_vel + gain_vel * input;
}

/* mjINT_IMPLICITFAST (MuJoCo "Computation / Numerical integration": implicit-in-velocity Euler with the velocity derivative of
   the smooth forces, Coriolis / centripetal terms dropped and the matrix kept symmetric):
     (M - h D) qacc* = qfrc_smooth + qfrc_constraint,   D = d(qfrc_passive + qfrc_actuator)/d(qvel)
                                                          = -diag(damping) - sum_t b_t J_t'J_t + sum_a s_a moment_a' moment_a
   with s_a = d force_a / d velocity_a (mjd_passive_vel / mjd_actuator_vel).  D is kept in the sparsity pattern of M (dof pairs
   on one kinematic chain -- mjData.qDeriv's pattern): a tendon whose two ends sit on different branches contributes its
   on-chain blocks only.  Dense Cholesky here (the oracle favours clarity). */
static void mmo_implicitfast(const mmo_model* m, mmo_data* d) {
  int nv = m->nv;
  real h = m->timestep;
  const int mk = smark(d);
  real* MM = salloc(d, nv * nv);
  real* Dm = salloc(d, nv * nv);
  mmo_full_m(m, d, MM);
  for (int i = 0; i < nv; i++) Dm[i * nv + i] -= MF(m, DOF_DAMPING)[i];
  for (int t = 0; t < m->ntendon; t++) {
    real b = MF(m, TENDON_DAMPING)[t];
    if (b == 0) continue;
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) Dm[i * nv + j] -= b * d->ten_J[t * nv + i] * d->ten_J[t * nv + j];
  }
  for (int a = 0; a < m->nu; a++) {
    real s = actuator_dforce_dvel(m, d, a);
    if (s == 0) continue;
    const real* mom = d->actuator_moment + (size_t)a * nv;
    for (int i = 0; i < nv; i++) if (mom[i] != 0) for (int j = 0; j < nv; j++) Dm[i * nv + j] += s * mom[i] * mom[j];
  }
  /* keep the pattern of M: (i, j) with j an ancestor-or-self of i (and its mirror) */
  for (int i = 0; i < nv; i++) {
    for (int j = 0; j < nv; j++) {
      int rel = 0;
      for (int k = i; k >= 0; k = MI(m, DOF_PARENTID)[k]) if (k == j) rel = 1;
      for (int k = j; k >= 0; k = MI(m, DOF_PARENTID)[k]) if (k == i) rel = 1;
      if (rel) MM[i * nv + j] -= h * Dm[i * nv + j];
    }
  }
  /* dense Cholesky MM = L L', solve */
  real* qacc = d->tmp_nv;
  for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  for (int j = 0; j < nv; j++) {
    real s = MM[j * nv + j];
    for (int k = 0; k < j; k++) s -= MM[j * nv + k] * MM[j * nv + k];
    real l = sqrt(fmax(s, MINVAL));
    MM[j * nv + j] = l;
    for (int i = j + 1; i < nv; i++) {
      real v = MM[i * nv + j];
      for (int k = 0; k < j; k++) v -= MM[i * nv + k] * MM[j * nv + k];
      MM[i * nv + j] = v / l;
    }
  }
  for (int i = 0; i < nv; i++) { real v = qacc[i]; for (int k = 0; k < i; k++) v -= MM[i * nv + k] * qacc[k]; qacc[i] = v / MM[i * nv + i]; }
  for (int i = nv - 1; i >= 0; i--) { real v = qacc[i]; for (int k = i + 1; k < nv; k++) v -= MM[k * nv + i] * qacc[k]; qacc[i] = v / MM[i * nv + i]; }
  srelease(d, mk);
  mmo_advance(m, d, qacc);
}

/* qpos <- qpos (+) hh * vel on the configuration manifold (mj_integratePos) */
static void integrate_pos(const mmo_model* m, real* qpos, const real* vel, real hh) {
  for (int j = 0; j < m->njnt; j++) {
    int type = MI(m, JNT_TYPE)[j], qa = MI(m, JNT_QPOSADR)[j], da = MI(m, JNT_DOFADR)[j];
    if (type == MM_JNT_HINGE || type == MM_JNT_SLIDE) { qpos[qa] += hh * vel[da]; continue; }
    if (type == MM_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += hh * vel[da + k];
      qa += 3; da += 3;
    }
    real w[3] = {vel[da], vel[da + 1], vel[da + 2]};
    real ang = hh * norm3(w);
    if (ang > MINVAL) {
      real ax[3] = {w[0], w[1], w[2]}, dq[4], qn[4];
      normalize3(ax);
      axisangle2quat(dq, ax, ang);
      quat_mul(qn, qpos + qa, dq);
      quat_normalize(qn);
      memcpy(qpos + qa, qn, sizeof(qn));
    }
  }
}

/* classical 4th-order Runge-Kutta over the state (qpos, qvel, act) with derivative (qvel, qacc, act_dot)
   (mj_RungeKutta, N = 4: a = {1/2, 1/2, 1}, b = {1/6, 1/3, 1/3, 1/6}); stage 0 is the forward pass mmo_step already ran.
   No implicit damping in this integrator; muscle activations are clamped to [0,1] at the final update only. */
static void mmo_rk4(const mmo_model* m, mmo_data* d) {
  int nv = m->nv, nq = m->nq, na = m->na;
  real h = m->timestep, t0 = d->time;
  static const real A[3] = {0.5, 0.5, 1.0}, B[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  const int mk = smark(d);
  real* q0 = salloc(d, nq); real* v0 = salloc(d, nv); real* a0 = salloc(d, na);
  real* vel = salloc(d, 4 * nv); real* acc = salloc(d, 4 * nv); real* adot = salloc(d, 4 * (na > 0 ? na : 1));
  memcpy(q0, d->qpos, sizeof(real) * nq); memcpy(v0, d->qvel, sizeof(real) * nv); memcpy(a0, d->act, sizeof(real) * na);
  for (int i = 0; i < 4; i++) {
    memcpy(vel + i * nv, d->qvel, sizeof(real) * nv); memcpy(acc + i * nv, d->qacc, sizeof(real) * nv);
    memcpy(adot + i * na, d->act_dot, sizeof(real) * na);
    if (i == 3) break;
    memcpy(d->qpos, q0, sizeof(real) * nq);
    integrate_pos(m, d->qpos, vel + i * nv, h * A[i]);
    for (int k = 0; k < nv; k++) d->qvel[k] = v0[k] + h * A[i] * acc[i * nv + k];
    for (int k = 0; k < na; k++) d->act[k] = a0[k] + h * A[i] * adot[i * na + k];
    d->time = t0 + h * A[i];
    mmo_forward(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(real) * nv);
  }
  real* dv = d->tmp_nv;
  memcpy(d->qpos, q0, sizeof(real) * nq);
  for (int k = 0; k < nv; k++) { dv[k] = 0; for (int i = 0; i < 4; i++) dv[k] += B[i] * vel[i * nv + k]; }
  integrate_pos(m, d->qpos, dv, h);
  for (int k = 0; k < nv; k++) { real s_ = 0; for (int i = 0; i < 4; i++) s_ += B[i] * acc[i * nv + k]; d->qvel[k] = v0[k] + h * s_; }
  for (int a = 0; a < m->nu; a++) {
    int aa = MI(m, ACT_ACTADR)[a];
    if (aa < 0) continue;
    real s_ = 0;
    for (int i = 0; i < 4; i++) s_ += B[i] * adot[i * na + aa];
    real x = a0[aa] + h * s_;
    if (MI(m, ACT_DYNTYPE)[a] == MM_DYN_MUSCLE) x = x < 0 ? 0 : (x > 1 ? 1 : x);
    d->act[aa] = x;
  }
  d->time = t0 + h;
  srelease(d, mk);
}

/* mj_step: forward + integrate, with MuJoCo's bad-state auto-reset semantics */
void mmo_step(const mmo_model* m, mmo_data* d) {
  /* mj_checkPos / mj_checkVel, then mj_checkAcc after the forward pass: warn, mj_resetData (which clears the controls as well: a
     caller that loops mj_step over one control vector -- Robot.step's frame_skip loop -- finishes the loop on zero input), carry on */
  if (bad_state(m, d, 0)) { mmo_reset(m, d); memset(d->ctrl, 0, sizeof(real) * m->nu); d->warn_bad |= 1; }
  mmo_forward(m, d);
  if (bad_state(m, d, 1)) { mmo_reset(m, d); memset(d->ctrl, 0, sizeof(real) * m->nu); d->warn_bad |= 1; mmo_forward(m, d); }
  memcpy(d->qacc_warmstart, d->qacc, sizeof(real) * m->nv);
  MMO_STAGE(MMO_ST_INTEG);
  if (m->integrator == MM_INT_RK4) mmo_rk4(m, d);
  else if (m->integrator == MM_INT_IMPLICITFAST) mmo_implicitfast(m, d);
  else mmo_euler(m, d);
  MMO_STAGE(MMO_ST_NONE);
  if (d->round_state_f32) {
    /* the fp32-STATE twin: all arithmetic in this file's precision, but the state handed from one substep to the next carries
       only a float's 24 bits -- the floor of any engine that stores its state in fp32 (tests/test_oracle_invariants.py) */
    for (int i = 0; i < m->nq; i++) d->qpos[i] = (real)(float)d->qpos[i];
    for (int i = 0; i < m->nv; i++) { d->qvel[i] = (real)(float)d->qvel[i]; d->qacc_warmstart[i] = (real)(float)d->qacc_warmstart[i]; }
    for (int i = 0; i < m->na; i++) d->act[i] = (real)(float)d->act[i];
  }
}

/* ---------------------------------------------------------- accessors */
typedef struct { const char* name; size_t off; int kind; } field_t;
#define FLD(n) {#n, offsetof(mmo_data, n), 0}
#include <stddef.h>
static const field_t kFields[] = {
    FLD(qpos), FLD(qvel), FLD(act), FLD(ctrl), FLD(qacc_warmstart), FLD(xpos), FLD(xquat), FLD(xmat),
    FLD(xipos), FLD(ximat), FLD(xanchor), FLD(xaxis), FLD(site_xpos), FLD(geom_xpos), FLD(geom_xmat),
    FLD(subtree_com), FLD(cinert), FLD(cdof), FLD(crb), FLD(ten_length), FLD(ten_J), FLD(actuator_length),
    FLD(actuator_moment), FLD(qM), FLD(qLD), FLD(qLDiagInv), FLD(ten_velocity), FLD(actuator_velocity),
    FLD(cvel), FLD(cdof_dot), FLD(qfrc_passive), FLD(qfrc_bias), FLD(act_dot), FLD(actuator_force),
    FLD(qfrc_actuator), FLD(qfrc_smooth), FLD(qacc_smooth), FLD(efc_J), FLD(efc_pos), FLD(efc_margin),
    FLD(efc_R), FLD(efc_D), FLD(efc_vel), FLD(efc_aref), FLD(efc_force), FLD(qfrc_constraint), FLD(qacc),
    FLD(cfrc), FLD(cacc), FLD(con_dist), FLD(con_pos), FLD(con_frame), FLD(efc_diagApprox), FLD(efc_floss)};

real* mmo_field(mmo_data* d, const char* name) {
  for (unsigned i = 0; i < sizeof(kFields) / sizeof(kFields[0]); i++)
    if (!strcmp(kFields[i].name, name)) return *(real**)((char*)d + kFields[i].off);
  return NULL;
}
void mmo_set_geom_size(mmo_data* d, int geom, double a, double b, double c) { d->gsize_id = geom; d->gsize_val[0] = a; d->gsize_val[1] = b; d->gsize_val[2] = c; }
void mmo_set_geom_type(mmo_data* d, int type) { d->gtype = type; }
void mmo_set_round_state_f32(mmo_data* d, int on) { d->round_state_f32 = on; }
void mmo_set_body_mass(mmo_data* d, int body, double mass) { d->bmass_id = body; d->bmass_val = mass; }
void mmo_set_body_pos(mmo_data* d, int body, double x, double y, double z) { d->bpos_id = body; d->bpos_val[0] = x; d->bpos_val[1] = y; d->bpos_val[2] = z; }
/* test hook: capsule-axis segment vs convex primitive in the primitive's frame -> signed distance, t, outward normal */
double mmo_test_seg_shape(int type, const double* size, const double* a, const double* u, double h, double* t, double* grad);
double mmo_time(const mmo_data* d) { return (double)d->time; }
void mmo_set_time(mmo_data* d, double t) { d->time = t; }
int mmo_nefc(const mmo_data* d) { return d->nefc; }
int mmo_ncon(const mmo_data* d) { return d->ncon; }
int mmo_solver_niter(const mmo_data* d) { return d->solver_niter; }
int mmo_warn(const mmo_data* d) { return d->warn_bad; }
const int* mmo_efc_type(const mmo_data* d) { return d->efc_type; }
const int* mmo_con_pair(const mmo_data* d) { return d->con_pair; }   /* PAIR_* entry of every detected contact */
int mmo_dim(const mmo_model* m, int which) { return MI(m, OPT_I)[which]; }

/* dense M (nv x nv) from the sparse layout, for tests */
void mmo_full_m(const mmo_model* m, const mmo_data* d, real* out) {
  int nv = m->nv;
  memset(out, 0, sizeof(real) * nv * nv);
  for (int i = 0; i < nv; i++) {
    int j = i, a = 0;
    while (j >= 0) { out[i * nv + j] = out[j * nv + i] = d->qM[MI(m, DOF_MADR)[i] + a]; j = MI(m, DOF_PARENTID)[j]; a++; }
  }
}

#ifndef MMO_REAL_EXTERNAL
double mmo_test_seg_shape(int type, const double* size, const double* a, const double* u, double h, double* t, double* grad) {
  return seg_shape(type, size, a, u, h, t, grad);
}
#endif
