/* myosim_model.h -- compiled musculoskeletal model blob ("MYOB") shared by the
 * fp64 CPU oracle (oracle/mmo_engine.c) and the HIP engine
 * (myosuite_amd/csrc/myosim_engine.hip).
 *
 * It plays the role that `mujoco.MjModel` plays at the reference's engine
 * boundary (reference call sites: myosuite/envs/env_base.py:72,75 build the
 * MjModel/MjData; myosuite/robot/robot.py:861 steps it).  The MuJoCo C library
 * itself is a third-party dependency absent from /root/reference
 * (pyproject.toml:31 `mujoco>=3.6,<3.7`), so field names and semantics follow
 * MuJoCo's published mjModel documentation; nothing here is copied code.
 *
 * Layout: one array of 32-bit words.
 *   word 0           MM_MAGIC
 *   word 1           MM_VERSION
 *   word 2           number of sections (MM_NSEC)
 *   word 3           total words in blob
 *   word 4+2*s       offset (in words, from blob start) of section s
 *   word 5+2*s       length (in words) of section s
 *   ...              section payloads (int32 or IEEE float32, see table)
 *
 * The section list below is an X-macro; the Python packer
 * (myosuite_amd/model/blob.py) parses THIS file, so the header is the single
 * source of truth for ids and dtypes.
 *   MM_SEC(NAME, 'i'|'f', words-per-element)   -- element count is implied by
 *   the owning dimension and checked by the packer.
 */
#ifndef MYOSIM_MODEL_H_
#define MYOSIM_MODEL_H_

#define MM_MAGIC   0x424F594D  /* "MYOB" little endian */
#define MM_VERSION 4
#define MM_HEADER_WORDS 4

/* ---- integer options / dimensions: indices into section OPT_I ------------ */
enum {
  MM_OI_NQ = 0, MM_OI_NV, MM_OI_NU, MM_OI_NA, MM_OI_NBODY, MM_OI_NJNT,
  MM_OI_NGEOM, MM_OI_NSITE, MM_OI_NTENDON, MM_OI_NWRAP, MM_OI_NEQ, MM_OI_NPAIR,
  MM_OI_NM,          /* non-zeros of tree-sparse inertia matrix            */
  MM_OI_NLEVEL,      /* depth levels of the body tree (world excluded)      */
  MM_OI_ITERATIONS,  /* Newton iterations cap  (MuJoCo opt.iterations)      */
  MM_OI_LS_ITERATIONS,
  MM_OI_INTEGRATOR,  /* 0 = semi-implicit Euler, 1 = RK4                    */
  MM_OI_EULERDAMP,   /* 1 = implicit joint damping in Euler                 */
  MM_OI_NJMAX,       /* upper bound on constraint rows (static)             */
  MM_OI_NTENJ,       /* upper bound on non-zeros of the tendon Jacobian     */
  MM_OI_NCONMAX,     /* upper bound on simultaneous contacts                */
  MM_OI_COUNT
};

/* ---- float options: indices into section OPT_F --------------------------- */
enum {
  MM_OF_TIMESTEP = 0, MM_OF_GRAV_X, MM_OF_GRAV_Y, MM_OF_GRAV_Z,
  MM_OF_TOLERANCE, MM_OF_LS_TOLERANCE, MM_OF_MEANINERTIA, MM_OF_IMPRATIO,
  MM_OF_COUNT
};

/* joint types (MuJoCo mjtJoint order) */
enum { MM_JNT_FREE = 0, MM_JNT_BALL = 1, MM_JNT_SLIDE = 2, MM_JNT_HINGE = 3 };
/* tendon path element types (MuJoCo mjtWrap) */
enum { MM_WRAP_NONE = 0, MM_WRAP_JOINT = 1, MM_WRAP_PULLEY = 2, MM_WRAP_SITE = 3,
       MM_WRAP_SPHERE = 4, MM_WRAP_CYLINDER = 5 };
/* geom types (MuJoCo mjtGeom order) */
enum { MM_GEOM_PLANE = 0, MM_GEOM_HFIELD = 1, MM_GEOM_SPHERE = 2, MM_GEOM_CAPSULE = 3,
       MM_GEOM_ELLIPSOID = 4, MM_GEOM_CYLINDER = 5, MM_GEOM_BOX = 6 };
/* actuator enums */
enum { MM_TRN_JOINT = 0, MM_TRN_TENDON = 3 };
enum { MM_DYN_NONE = 0, MM_DYN_INTEGRATOR = 1, MM_DYN_FILTER = 2, MM_DYN_MUSCLE = 4 };   /* mjtDyn; filter: act_dot = (ctrl - act) / dynprm0 */
enum { MM_GAIN_FIXED = 0, MM_GAIN_MUSCLE = 2 };
enum { MM_BIAS_NONE = 0, MM_BIAS_AFFINE = 1, MM_BIAS_MUSCLE = 2 };   /* affine: biasprm0 + biasprm1*length + biasprm2*velocity */
/* equality types */
enum { MM_EQ_JOINT = 2 };
/* mjtIntegrator values carried in MM_OI_INTEGRATOR */
enum { MM_INT_EULER = 0, MM_INT_RK4 = 1, MM_INT_IMPLICITFAST = 3 };   /* mjtIntegrator values (2 = the full `implicit`, not implemented) */
/* constraint row types (oracle + engine internal) */
enum { MM_CON_EQUALITY = 0, MM_CON_LIMIT_JOINT = 1, MM_CON_LIMIT_TENDON = 2,
       MM_CON_CONTACT = 3, MM_CON_FRICTION_DOF = 4 };

#define MM_MINVAL 1e-15  /* MuJoCo mjMINVAL */
#define MM_MAX_PAIRS 256 /* longest explicit contact-pair list (PAIR_* sections) the HIP engine sweeps: chunks of one pair per lane */

/* ---- sections ------------------------------------------------------------ */
#define MM_SECTIONS(MM_SEC)                                                      \
  MM_SEC(OPT_I,            'i', 1)   /* [MM_OI_COUNT]                        */ \
  MM_SEC(OPT_F,            'f', 1)   /* [MM_OF_COUNT]                        */ \
  /* bodies, body 0 = world */                                                   \
  MM_SEC(BODY_PARENT,      'i', 1)                                               \
  MM_SEC(BODY_ROOTID,      'i', 1)                                               \
  MM_SEC(BODY_JNTADR,      'i', 1)   /* -1 if none */                            \
  MM_SEC(BODY_JNTNUM,      'i', 1)                                               \
  MM_SEC(BODY_DOFADR,      'i', 1)   /* -1 if none */                            \
  MM_SEC(BODY_DOFNUM,      'i', 1)                                               \
  MM_SEC(BODY_POS,         'f', 3)                                               \
  MM_SEC(BODY_QUAT,        'f', 4)                                               \
  MM_SEC(BODY_IPOS,        'f', 3)                                               \
  MM_SEC(BODY_IQUAT,       'f', 4)                                               \
  MM_SEC(BODY_MASS,        'f', 1)                                               \
  MM_SEC(BODY_INERTIA,     'f', 3)                                               \
  MM_SEC(BODY_INVWEIGHT0,  'f', 2)                                               \
  /* joints */                                                                   \
  MM_SEC(JNT_TYPE,         'i', 1)                                               \
  MM_SEC(JNT_BODYID,       'i', 1)                                               \
  MM_SEC(JNT_QPOSADR,      'i', 1)                                               \
  MM_SEC(JNT_DOFADR,       'i', 1)                                               \
  MM_SEC(JNT_LIMITED,      'i', 1)                                               \
  MM_SEC(JNT_POS,          'f', 3)                                               \
  MM_SEC(JNT_AXIS,         'f', 3)                                               \
  MM_SEC(JNT_STIFFNESS,    'f', 1)                                               \
  MM_SEC(JNT_RANGE,        'f', 2)                                               \
  MM_SEC(JNT_MARGIN,       'f', 1)                                               \
  MM_SEC(JNT_SOLREF,       'f', 2)                                               \
  MM_SEC(JNT_SOLIMP,       'f', 5)                                               \
  /* degrees of freedom */                                                       \
  MM_SEC(DOF_BODYID,       'i', 1)                                               \
  MM_SEC(DOF_JNTID,        'i', 1)                                               \
  MM_SEC(DOF_PARENTID,     'i', 1)   /* -1 at tree root */                       \
  MM_SEC(DOF_MADR,         'i', 1)                                               \
  MM_SEC(DOF_DAMPING,      'f', 1)                                               \
  MM_SEC(DOF_ARMATURE,     'f', 1)                                               \
  MM_SEC(DOF_INVWEIGHT0,   'f', 1)                                               \
  MM_SEC(DOF_FRICTIONLOSS, 'f', 1)   /* dry friction (joint frictionloss)    */ \
  MM_SEC(DOF_SOLREF,       'f', 2)   /* solreffriction                       */ \
  MM_SEC(DOF_SOLIMP,       'f', 5)   /* solimpfriction                       */ \
  MM_SEC(QPOS0,            'f', 1)   /* [nq] */                                  \
  MM_SEC(QPOS_SPRING,      'f', 1)   /* [nq] */                                  \
  /* sites */                                                                    \
  MM_SEC(SITE_BODYID,      'i', 1)                                               \
  MM_SEC(SITE_POS,         'f', 3)                                               \
  /* geoms (wrapping obstacles and collision shapes) */                          \
  MM_SEC(GEOM_TYPE,        'i', 1)                                               \
  MM_SEC(GEOM_BODYID,      'i', 1)                                               \
  MM_SEC(GEOM_POS,         'f', 3)                                               \
  MM_SEC(GEOM_QUAT,        'f', 4)                                               \
  MM_SEC(GEOM_SIZE,        'f', 3)                                               \
  /* tendons */                                                                  \
  MM_SEC(TENDON_ADR,       'i', 1)                                               \
  MM_SEC(TENDON_NUM,       'i', 1)                                               \
  MM_SEC(TENDON_LIMITED,   'i', 1)                                               \
  MM_SEC(TENDON_RANGE,     'f', 2)                                               \
  MM_SEC(TENDON_MARGIN,    'f', 1)                                               \
  MM_SEC(TENDON_STIFFNESS, 'f', 1)                                               \
  MM_SEC(TENDON_DAMPING,   'f', 1)                                               \
  MM_SEC(TENDON_LENGTHSPRING,'f', 2)                                             \
  MM_SEC(TENDON_SOLREF,    'f', 2)                                               \
  MM_SEC(TENDON_SOLIMP,    'f', 5)                                               \
  MM_SEC(TENDON_INVWEIGHT0,'f', 1)                                               \
  MM_SEC(WRAP_TYPE,        'i', 1)                                               \
  MM_SEC(WRAP_OBJID,       'i', 1)                                               \
  MM_SEC(WRAP_PRM,         'f', 1)   /* divisor | side-site id | joint coef */   \
  /* actuators */                                                                \
  MM_SEC(ACT_TRNTYPE,      'i', 1)                                               \
  MM_SEC(ACT_TRNID,        'i', 1)                                               \
  MM_SEC(ACT_DYNTYPE,      'i', 1)                                               \
  MM_SEC(ACT_GAINTYPE,     'i', 1)                                               \
  MM_SEC(ACT_BIASTYPE,     'i', 1)                                               \
  MM_SEC(ACT_ACTADR,       'i', 1)   /* -1 if stateless */                       \
  MM_SEC(ACT_CTRLLIMITED,  'i', 1)                                               \
  MM_SEC(ACT_FORCELIMITED, 'i', 1)                                               \
  MM_SEC(ACT_GEAR,         'f', 1)                                               \
  MM_SEC(ACT_DYNPRM,       'f', 3)                                               \
  MM_SEC(ACT_GAINPRM,      'f', 9)                                               \
  MM_SEC(ACT_BIASPRM,      'f', 9)                                               \
  MM_SEC(ACT_CTRLRANGE,    'f', 2)                                               \
  MM_SEC(ACT_FORCERANGE,   'f', 2)                                               \
  MM_SEC(ACT_LENGTHRANGE,  'f', 2)                                               \
  MM_SEC(ACT_ACC0,         'f', 1)                                               \
  /* equality constraints (joint coupling polynomials) */                        \
  MM_SEC(EQ_TYPE,          'i', 1)                                               \
  MM_SEC(EQ_OBJ1ID,        'i', 1)                                               \
  MM_SEC(EQ_OBJ2ID,        'i', 1)   /* -1: couple to constant */                \
  MM_SEC(EQ_DATA,          'f', 5)   /* polycoef */                              \
  MM_SEC(EQ_SOLREF,        'f', 2)                                               \
  MM_SEC(EQ_SOLIMP,        'f', 5)                                               \
  /* static candidate contact pairs with pre-mixed parameters; type(geom1) <= type(geom2).  An entry yields at most two   \
     contacts: a pair whose collider can return four (plane-box, plane-cylinder) takes TWO consecutive identical entries, \
     entry k of the run keeping contacts 2k, 2k+1 of the collider's order (oracle/mmo_collision.inc) */                   \
  MM_SEC(PAIR_GEOM1,       'i', 1)                                               \
  MM_SEC(PAIR_GEOM2,       'i', 1)                                               \
  MM_SEC(PAIR_CONDIM,      'i', 1)   /* 1 | 3 | 4: pyramidal cone, 1 / 4 / 6 rows */  \
  MM_SEC(PAIR_FRICTION,    'f', 3)   /* slide, spin, roll */                     \
  MM_SEC(PAIR_MARGIN,      'f', 1)                                               \
  MM_SEC(PAIR_GAP,         'f', 1)                                               \
  MM_SEC(PAIR_SOLREF,      'f', 2)                                               \
  MM_SEC(PAIR_SOLIMP,      'f', 5)                                               \
  /* scheduling helpers for the wave-cooperative HIP engine */                   \
  MM_SEC(LEVEL_ADR,        'i', 1)   /* [nlevel+1] into LEVEL_BODY */            \
  MM_SEC(LEVEL_BODY,       'i', 1)   /* [nbody-1] bodies sorted by depth */      \
  MM_SEC(DOF_LEVEL_ADR,    'i', 1)   /* [ndoflevel+1] into DOF_LEVEL_DOF */      \
  MM_SEC(DOF_LEVEL_DOF,    'i', 1)   /* dofs sorted by depth in dof tree */      \
  MM_SEC(TENJ_ADR,         'i', 1)   /* [ntendon+1] row starts of sparse ten_J */\
  MM_SEC(TENJ_DOF,         'i', 1)   /* [NTENJ] dof index per non-zero */

enum {
#define MM_SEC_ENUM(NAME, T, W) MM_SEC_##NAME,
  MM_SECTIONS(MM_SEC_ENUM)
#undef MM_SEC_ENUM
  MM_NSEC
};

#endif /* MYOSIM_MODEL_H_ */
