"""ModelSpec: authoring + host-side compilation of musculoskeletal models.

Plays the role of ``mujoco.MjSpec`` + ``.compile()`` at the reference boundary
(myosuite/envs/env_base.py:72,96-106).  The reference's real models live in the
``myo_sim`` submodule, which is EMPTY in /root/reference (.gitmodules:1-3), so
models are authored programmatically (see synth.py) with the dimensions that
in-repo evidence pins.

``compile()`` is setup-time host logic (numpy): it lays the tree out as flat SoA
arrays (MuJoCo mjModel field semantics), derives the compile-time constants the
engine needs (``dof_invweight0``, ``body_invweight0``, ``tendon_invweight0``,
``stat.meaninertia``, muscle ``lengthrange``/``acc0``) and the scheduling tables
of the wave-cooperative HIP engine (body levels, dof levels, sparse tendon
Jacobian pattern).  Nothing here runs per step; the per-step pipeline is the HIP
engine (myosuite_amd/csrc) and, for checking only, the fp64 oracle (oracle/).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import blob as _blob
from . import kin_np as K

C = _blob.C

DEFAULT_SOLREF = (0.02, 1.0)
DEFAULT_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)
# MuJoCo default muscle parameters: range(2) force scale lmin lmax vmax fpmax fvmax
DEFAULT_MUSCLE_PRM = (0.75, 1.05, -1.0, 200.0, 0.5, 1.6, 1.5, 1.3, 1.2)


@dataclasses.dataclass
class _Body:
    name: str
    parent: int
    pos: np.ndarray
    quat: np.ndarray
    mass: float
    ipos: np.ndarray
    iquat: np.ndarray
    inertia: np.ndarray
    joints: List[int] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class _Joint:
    name: str
    body: int
    type: int
    pos: np.ndarray
    axis: np.ndarray
    limited: bool
    range: Tuple[float, float]
    stiffness: float
    damping: float
    armature: float
    ref: float
    springref: float
    margin: float
    solref: Tuple[float, float]
    solimp: Tuple[float, ...]
    frictionloss: float = 0.0
    solreffriction: Tuple[float, float] = DEFAULT_SOLREF
    solimpfriction: Tuple[float, ...] = DEFAULT_SOLIMP


@dataclasses.dataclass
class _Tendon:
    name: str
    path: list
    limited: bool
    range: Tuple[float, float]
    margin: float
    stiffness: float
    damping: float
    springlength: Tuple[float, float]
    solref: Tuple[float, float]
    solimp: Tuple[float, ...]


@dataclasses.dataclass
class _Actuator:
    name: str
    trntype: int
    target: str
    gear: float
    dyntype: int
    gaintype: int
    biastype: int
    dynprm: Tuple[float, float, float]
    gainprm: Tuple[float, ...]
    biasprm: Tuple[float, ...]
    ctrllimited: bool
    ctrlrange: Tuple[float, float]
    forcelimited: bool
    forcerange: Tuple[float, float]
    lengthrange: Optional[Tuple[float, float]]


class CompiledModel:
    """Flat arrays (dict keyed by section NAME) + name maps + packed blob."""

    def __init__(self, name: str, arrays: Dict[str, np.ndarray], names: Dict[str, Dict[str, int]]):
        self.name = name
        self.arrays = arrays
        self.names = names
        self.blob = _blob.pack(arrays)
        oi = arrays["OPT_I"]
        self.nq = int(oi[C["MM_OI_NQ"]]); self.nv = int(oi[C["MM_OI_NV"]])
        self.nu = int(oi[C["MM_OI_NU"]]); self.na = int(oi[C["MM_OI_NA"]])
        self.nbody = int(oi[C["MM_OI_NBODY"]]); self.njnt = int(oi[C["MM_OI_NJNT"]])
        self.ngeom = int(oi[C["MM_OI_NGEOM"]]); self.nsite = int(oi[C["MM_OI_NSITE"]])
        self.ntendon = int(oi[C["MM_OI_NTENDON"]]); self.nwrap = int(oi[C["MM_OI_NWRAP"]])
        self.neq = int(oi[C["MM_OI_NEQ"]]); self.npair = int(oi[C["MM_OI_NPAIR"]])
        self.nM = int(oi[C["MM_OI_NM"]]); self.njmax = int(oi[C["MM_OI_NJMAX"]])
        self.ntenJ = int(oi[C["MM_OI_NTENJ"]]); self.nconmax = int(oi[C["MM_OI_NCONMAX"]])
        self.timestep = float(arrays["OPT_F"][C["MM_OF_TIMESTEP"]])

    # convenience views used by the env layer
    @property
    def jnt_range(self) -> np.ndarray:
        return self.arrays["JNT_RANGE"].reshape(-1, 2)

    @property
    def qpos0(self) -> np.ndarray:
        return self.arrays["QPOS0"]

    def joint_id(self, name: str) -> int:
        return self.names["joint"][name]

    def site_id(self, name: str) -> int:
        return self.names["site"][name]

    def body_id(self, name: str) -> int:
        return self.names["body"][name]

    def hash(self) -> str:
        import hashlib
        return hashlib.sha256(self.blob.tobytes()).hexdigest()[:16]


def _v(x, n):
    a = np.asarray(x, dtype=np.float64).reshape(-1)
    assert a.size == n, f"expected {n} values, got {a.size}"
    return a


class ModelSpec:
    def __init__(self, name: str, timestep: float = 0.002, gravity=(0.0, 0.0, -9.81),
                 tolerance: float = 1e-8, iterations: int = 100, ls_iterations: int = 50,
                 ls_tolerance: float = 0.01, integrator: int = 0, eulerdamp: bool = True,
                 nconmax: int = 0):
        self.name = name
        self.timestep = timestep
        self.gravity = _v(gravity, 3)
        self.tolerance = tolerance
        self.iterations = iterations
        self.ls_iterations = ls_iterations
        self.ls_tolerance = ls_tolerance
        self.integrator = integrator
        self.eulerdamp = eulerdamp
        self.nconmax = nconmax
        self.njmax = 0          # 0: derived from the row kinds (limits + nconmax contacts); > 0: explicit row bound (mjModel.njmax)
        self.bodies: List[_Body] = [_Body("world", -1, np.zeros(3), np.array([1., 0, 0, 0]), 0.0,
                                          np.zeros(3), np.array([1., 0, 0, 0]), np.zeros(3))]
        self.joints: List[_Joint] = []
        self.sites: List[Tuple[str, int, np.ndarray]] = []
        self.geoms: List[dict] = []
        self.tendons: List[_Tendon] = []
        self.actuators: List[_Actuator] = []
        self.equalities: List[dict] = []
        self.pairs: List[dict] = []
        self._bname: Dict[str, int] = {"world": 0}
        self._sname: Dict[str, int] = {}
        self._gname: Dict[str, int] = {}
        self._jname: Dict[str, int] = {}
        self._tname: Dict[str, int] = {}

    # ---------------------------------------------------------------- authoring
    def add_body(self, name, parent="world", pos=(0, 0, 0), quat=(1, 0, 0, 0), mass=0.0,
                 ipos=(0, 0, 0), inertia=(0, 0, 0), iquat=(1, 0, 0, 0)) -> int:
        assert name not in self._bname, name
        p = self._bname[parent]
        q = _v(quat, 4); q = q / np.linalg.norm(q)
        iq = _v(iquat, 4); iq = iq / np.linalg.norm(iq)
        self.bodies.append(_Body(name, p, _v(pos, 3), q, float(mass), _v(ipos, 3), iq, _v(inertia, 3)))
        self._bname[name] = len(self.bodies) - 1
        return len(self.bodies) - 1

    def add_joint(self, name, body, type="hinge", pos=(0, 0, 0), axis=(0, 0, 1), range=None,
                  stiffness=0.0, damping=0.0, armature=0.0, ref=0.0, springref=0.0, margin=0.0,
                  solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP, frictionloss=0.0, solreffriction=DEFAULT_SOLREF,
                  solimpfriction=DEFAULT_SOLIMP) -> int:
        assert name not in self._jname, name
        b = self._bname[body]
        # MuJoCo requires joints of a body to be contiguous: enforce authoring order
        assert b == len(self.bodies) - 1 or not any(j.body > b for j in self.joints), \
            "add joints right after their body"
        t = {"free": C["MM_JNT_FREE"], "ball": C["MM_JNT_BALL"], "slide": C["MM_JNT_SLIDE"],
             "hinge": C["MM_JNT_HINGE"]}[type]
        ax = _v(axis, 3)
        if t in (C["MM_JNT_SLIDE"], C["MM_JNT_HINGE"]):
            ax = ax / np.linalg.norm(ax)
        lim = range is not None
        rng = (float(range[0]), float(range[1])) if lim else (0.0, 0.0)
        self.joints.append(_Joint(name, b, t, _v(pos, 3), ax, lim, rng, float(stiffness), float(damping),
                                  float(armature), float(ref), float(springref), float(margin),
                                  tuple(solref), tuple(solimp), float(frictionloss), tuple(solreffriction),
                                  tuple(solimpfriction)))
        jid = len(self.joints) - 1
        self.bodies[b].joints.append(jid)
        self._jname[name] = jid
        return jid

    def add_site(self, name, body, pos) -> int:
        assert name not in self._sname, name
        self.sites.append((name, self._bname[body], _v(pos, 3)))
        self._sname[name] = len(self.sites) - 1
        return len(self.sites) - 1

    def add_geom(self, name, body, type, size, pos=(0, 0, 0), quat=(1, 0, 0, 0)) -> int:
        assert name not in self._gname, name
        t = {"plane": 0, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6}[type]
        s = np.zeros(3); sz = np.asarray(size, dtype=np.float64).reshape(-1); s[:sz.size] = sz
        q = _v(quat, 4); q = q / np.linalg.norm(q)
        self.geoms.append(dict(name=name, body=self._bname[body], type=t, size=s, pos=_v(pos, 3), quat=q))
        self._gname[name] = len(self.geoms) - 1
        return len(self.geoms) - 1

    def add_tendon(self, name, path, limited=False, range=(0.0, 0.0), margin=0.0, stiffness=0.0,
                   damping=0.0, springlength=(-1.0, -1.0), solref=DEFAULT_SOLREF,
                   solimp=DEFAULT_SOLIMP) -> int:
        """path: list of ('site', name) | ('sphere'|'cylinder', geom, sidesite|None) | ('pulley', divisor)"""
        assert name not in self._tname, name
        self.tendons.append(_Tendon(name, list(path), bool(limited), tuple(range), float(margin),
                                    float(stiffness), float(damping), tuple(springlength),
                                    tuple(solref), tuple(solimp)))
        self._tname[name] = len(self.tendons) - 1
        return len(self.tendons) - 1

    def add_muscle(self, name, tendon, force, range=(0.75, 1.05), scale=200.0, lmin=0.5, lmax=1.6,
                   vmax=1.5, fpmax=1.3, fvmax=1.2, tau=(0.01, 0.04, 0.0), lengthrange=None,
                   gear=1.0) -> int:
        prm = (range[0], range[1], force, scale, lmin, lmax, vmax, fpmax, fvmax)
        self.actuators.append(_Actuator(name, C["MM_TRN_TENDON"], tendon, float(gear), C["MM_DYN_MUSCLE"],
                                        C["MM_GAIN_MUSCLE"], C["MM_BIAS_MUSCLE"], tuple(tau), prm, prm,
                                        True, (0.0, 1.0), False, (0.0, 0.0), lengthrange))
        return len(self.actuators) - 1

    def add_motor(self, name, joint, gear=1.0, ctrlrange=None) -> int:
        z9 = (1.0,) + (0.0,) * 8
        self.actuators.append(_Actuator(name, C["MM_TRN_JOINT"], joint, float(gear), C["MM_DYN_NONE"],
                                        C["MM_GAIN_FIXED"], C["MM_BIAS_NONE"], (1.0, 0.0, 0.0), z9,
                                        (0.0,) * 9, ctrlrange is not None,
                                        tuple(ctrlrange) if ctrlrange else (0.0, 0.0), False, (0.0, 0.0), None))
        return len(self.actuators) - 1

    def add_general(self, name, joint=None, tendon=None, gear=1.0, gainprm=(1.0,), biasprm=(0.0, 0.0, 0.0), ctrlrange=None,
                    forcerange=None, dyntype="none", dynprm=(1.0, 0.0, 0.0)) -> int:
        """<general> / <position> / <velocity> without activation dynamics: force = gainprm0 * ctrl + biasprm0 +
        biasprm1 * length + biasprm2 * velocity (position servo kp: gainprm0 = kp, biasprm1 = -kp; velocity servo kv:
        gainprm0 = kv, biasprm2 = -kv)."""
        assert (joint is None) != (tendon is None)
        z9 = (0.0,) * 9
        gp = (tuple(float(x) for x in gainprm) + z9)[:9]; bp = (tuple(float(x) for x in biasprm) + z9)[:9]
        affine = any(x != 0.0 for x in bp)
        self.actuators.append(_Actuator(name, C["MM_TRN_JOINT"] if joint is not None else C["MM_TRN_TENDON"],
                                        joint if joint is not None else tendon, float(gear),
                                        {"none": C["MM_DYN_NONE"], "integrator": C["MM_DYN_INTEGRATOR"], "filter": C["MM_DYN_FILTER"]}[dyntype],
                                        C["MM_GAIN_FIXED"], C["MM_BIAS_AFFINE"] if affine else C["MM_BIAS_NONE"],
                                        (tuple(float(x) for x in dynprm) + (0.0, 0.0))[:3], gp, bp,
                                        ctrlrange is not None, tuple(ctrlrange) if ctrlrange else (0.0, 0.0),
                                        forcerange is not None, tuple(forcerange) if forcerange else (0.0, 0.0), None))
        return len(self.actuators) - 1

    def add_equality_joint(self, joint1, joint2, polycoef, solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP):
        pc = np.zeros(5); pc[:len(polycoef)] = polycoef
        self.equalities.append(dict(j1=joint1, j2=joint2, data=pc, solref=tuple(solref), solimp=tuple(solimp)))

    def add_contact_pair(self, geom1, geom2, condim=3, friction=(1.0, 0.005, 0.0001), margin=0.0, gap=0.0,
                         solref=DEFAULT_SOLREF, solimp=DEFAULT_SOLIMP):
        self.pairs.append(dict(g1=geom1, g2=geom2, condim=int(condim), friction=tuple(friction),
                               margin=float(margin), gap=float(gap), solref=tuple(solref), solimp=tuple(solimp)))

    # ---------------------------------------------------------------- compile
    def compile(self, lengthrange_samples: int = 4096, seed: int = 0) -> CompiledModel:
        nbody = len(self.bodies)
        for i, b in enumerate(self.bodies[1:], start=1):
            assert b.parent < i
        # joints are already contiguous per body in authoring order; verify & index
        jnt_order = []
        for b in self.bodies:
            jnt_order.extend(b.joints)
        assert jnt_order == list(range(len(self.joints))), "joints must be authored in body order"
        njnt = len(self.joints)
        qposadr, dofadr = [], []
        nq = nv = 0
        for j in self.joints:
            qposadr.append(nq); dofadr.append(nv)
            dq, dv = {0: (7, 6), 1: (4, 3), 2: (1, 1), 3: (1, 1)}[j.type]
            nq += dq; nv += dv
        A: Dict[str, np.ndarray] = {}
        i32, f32 = np.int32, np.float32
        body_jntadr = np.full(nbody, -1, i32); body_jntnum = np.zeros(nbody, i32)
        body_dofadr = np.full(nbody, -1, i32); body_dofnum = np.zeros(nbody, i32)
        for bi, b in enumerate(self.bodies):
            if b.joints:
                body_jntadr[bi] = b.joints[0]; body_jntnum[bi] = len(b.joints)
                body_dofadr[bi] = dofadr[b.joints[0]]
                body_dofnum[bi] = sum({0: 6, 1: 3, 2: 1, 3: 1}[self.joints[j].type] for j in b.joints)
        parent = np.array([b.parent for b in self.bodies], i32)
        rootid = np.zeros(nbody, i32)
        for i in range(1, nbody):
            rootid[i] = i if parent[i] == 0 else rootid[parent[i]]
        A["BODY_PARENT"] = parent; A["BODY_ROOTID"] = rootid
        A["BODY_JNTADR"] = body_jntadr; A["BODY_JNTNUM"] = body_jntnum
        A["BODY_DOFADR"] = body_dofadr; A["BODY_DOFNUM"] = body_dofnum
        A["BODY_POS"] = np.array([b.pos for b in self.bodies], f32)
        A["BODY_QUAT"] = np.array([b.quat for b in self.bodies], f32)
        A["BODY_IPOS"] = np.array([b.ipos for b in self.bodies], f32)
        A["BODY_IQUAT"] = np.array([b.iquat for b in self.bodies], f32)
        A["BODY_MASS"] = np.array([b.mass for b in self.bodies], f32)
        A["BODY_INERTIA"] = np.array([b.inertia for b in self.bodies], f32)

        A["JNT_TYPE"] = np.array([j.type for j in self.joints], i32)
        A["JNT_BODYID"] = np.array([j.body for j in self.joints], i32)
        A["JNT_QPOSADR"] = np.array(qposadr, i32)
        A["JNT_DOFADR"] = np.array(dofadr, i32)
        A["JNT_LIMITED"] = np.array([int(j.limited) for j in self.joints], i32)
        A["JNT_POS"] = np.array([j.pos for j in self.joints], f32).reshape(njnt, 3)
        A["JNT_AXIS"] = np.array([j.axis for j in self.joints], f32).reshape(njnt, 3)
        A["JNT_STIFFNESS"] = np.array([j.stiffness for j in self.joints], f32)
        A["JNT_RANGE"] = np.array([j.range for j in self.joints], f32).reshape(njnt, 2)
        A["JNT_MARGIN"] = np.array([j.margin for j in self.joints], f32)
        A["JNT_SOLREF"] = np.array([j.solref for j in self.joints], f32).reshape(njnt, 2)
        A["JNT_SOLIMP"] = np.array([j.solimp for j in self.joints], f32).reshape(njnt, 5)

        dof_bodyid = np.zeros(nv, i32); dof_jntid = np.zeros(nv, i32)
        dof_damping = np.zeros(nv, f32); dof_armature = np.zeros(nv, f32)
        dof_floss = np.zeros(nv, f32); dof_solref = np.zeros((nv, 2), f32); dof_solimp = np.zeros((nv, 5), f32)
        qpos0 = np.zeros(nq, f32); qpos_spring = np.zeros(nq, f32)
        for ji, j in enumerate(self.joints):
            dv = {0: 6, 1: 3, 2: 1, 3: 1}[j.type]
            for k in range(dv):
                d = dofadr[ji] + k
                dof_bodyid[d] = j.body; dof_jntid[d] = ji
                dof_damping[d] = j.damping; dof_armature[d] = j.armature
                dof_floss[d] = j.frictionloss; dof_solref[d] = j.solreffriction; dof_solimp[d] = j.solimpfriction
            qa = qposadr[ji]
            if j.type == C["MM_JNT_FREE"]:
                qpos0[qa:qa + 3] = self.bodies[j.body].pos; qpos0[qa + 3:qa + 7] = self.bodies[j.body].quat
                qpos_spring[qa:qa + 7] = qpos0[qa:qa + 7]
            elif j.type == C["MM_JNT_BALL"]:
                qpos0[qa:qa + 4] = (1, 0, 0, 0); qpos_spring[qa:qa + 4] = (1, 0, 0, 0)
            else:
                qpos0[qa] = j.ref; qpos_spring[qa] = j.springref
        # dof tree: previous dof in same body, else last dof of nearest ancestor with dofs
        dof_parent = np.full(nv, -1, i32)
        last_dof_of_body = np.full(nbody, -1, i32)
        for bi in range(1, nbody):
            anc = last_dof_of_body[parent[bi]] if parent[bi] >= 0 else -1
            if body_dofnum[bi] > 0:
                for k in range(body_dofnum[bi]):
                    d = body_dofadr[bi] + k
                    dof_parent[d] = anc if k == 0 else d - 1
                last_dof_of_body[bi] = body_dofadr[bi] + body_dofnum[bi] - 1
            else:
                last_dof_of_body[bi] = anc
        dof_madr = np.zeros(nv, i32); nM = 0
        dof_depth = np.zeros(nv, i32)
        for d in range(nv):
            dof_madr[d] = nM
            k = d; cnt = 0
            while k >= 0:
                cnt += 1; k = dof_parent[k]
            nM += cnt; dof_depth[d] = cnt - 1
        A["DOF_BODYID"] = dof_bodyid; A["DOF_JNTID"] = dof_jntid
        A["DOF_PARENTID"] = dof_parent; A["DOF_MADR"] = dof_madr
        A["DOF_DAMPING"] = dof_damping; A["DOF_ARMATURE"] = dof_armature
        A["DOF_FRICTIONLOSS"] = dof_floss; A["DOF_SOLREF"] = dof_solref; A["DOF_SOLIMP"] = dof_solimp
        A["QPOS0"] = qpos0; A["QPOS_SPRING"] = qpos_spring

        nsite = len(self.sites)
        A["SITE_BODYID"] = np.array([s[1] for s in self.sites], i32)
        A["SITE_POS"] = np.array([s[2] for s in self.sites], f32).reshape(nsite, 3)
        ngeom = len(self.geoms)
        A["GEOM_TYPE"] = np.array([g["type"] for g in self.geoms], i32)
        A["GEOM_BODYID"] = np.array([g["body"] for g in self.geoms], i32)
        A["GEOM_POS"] = np.array([g["pos"] for g in self.geoms], f32).reshape(ngeom, 3)
        A["GEOM_QUAT"] = np.array([g["quat"] for g in self.geoms], f32).reshape(ngeom, 4)
        A["GEOM_SIZE"] = np.array([g["size"] for g in self.geoms], f32).reshape(ngeom, 3)

        # tendons / wrap path
        wt, wo, wp = [], [], []
        tadr, tnum = [], []
        for t in self.tendons:
            tadr.append(len(wt))
            for el in t.path:
                kind = el[0]
                if kind == "site":
                    wt.append(C["MM_WRAP_SITE"]); wo.append(self._sname[el[1]]); wp.append(0.0)
                elif kind in ("sphere", "cylinder"):
                    g = self._gname[el[1]]
                    gt = self.geoms[g]["type"]
                    assert gt == (2 if kind == "sphere" else 5), "wrap geom type mismatch"
                    wt.append(C["MM_WRAP_SPHERE"] if kind == "sphere" else C["MM_WRAP_CYLINDER"])
                    wo.append(g)
                    side = el[2] if len(el) > 2 else None
                    wp.append(float(self._sname[side]) if side is not None else -1.0)
                elif kind == "pulley":
                    wt.append(C["MM_WRAP_PULLEY"]); wo.append(-1); wp.append(float(el[1]))
                elif kind == "joint":
                    wt.append(C["MM_WRAP_JOINT"]); wo.append(self._jname[el[1]]); wp.append(float(el[2]))
                else:
                    raise ValueError(kind)
            tnum.append(len(wt) - tadr[-1])
        ntendon = len(self.tendons); nwrap = len(wt)
        A["TENDON_ADR"] = np.array(tadr, i32); A["TENDON_NUM"] = np.array(tnum, i32)
        A["TENDON_LIMITED"] = np.array([int(t.limited) for t in self.tendons], i32)
        A["TENDON_RANGE"] = np.array([t.range for t in self.tendons], f32).reshape(ntendon, 2)
        A["TENDON_MARGIN"] = np.array([t.margin for t in self.tendons], f32)
        A["TENDON_STIFFNESS"] = np.array([t.stiffness for t in self.tendons], f32)
        A["TENDON_DAMPING"] = np.array([t.damping for t in self.tendons], f32)
        A["TENDON_SOLREF"] = np.array([t.solref for t in self.tendons], f32).reshape(ntendon, 2)
        A["TENDON_SOLIMP"] = np.array([t.solimp for t in self.tendons], f32).reshape(ntendon, 5)
        A["WRAP_TYPE"] = np.array(wt, i32); A["WRAP_OBJID"] = np.array(wo, i32)
        A["WRAP_PRM"] = np.array(wp, f32)

        # actuators
        nu = len(self.actuators)
        na = 0
        actadr = []
        for a in self.actuators:
            if a.dyntype != C["MM_DYN_NONE"]:
                actadr.append(na); na += 1
            else:
                actadr.append(-1)
        trnid = []
        for a in self.actuators:
            trnid.append(self._tname[a.target] if a.trntype == C["MM_TRN_TENDON"] else self._jname[a.target])
        A["ACT_TRNTYPE"] = np.array([a.trntype for a in self.actuators], i32)
        A["ACT_TRNID"] = np.array(trnid, i32)
        A["ACT_DYNTYPE"] = np.array([a.dyntype for a in self.actuators], i32)
        A["ACT_GAINTYPE"] = np.array([a.gaintype for a in self.actuators], i32)
        A["ACT_BIASTYPE"] = np.array([a.biastype for a in self.actuators], i32)
        A["ACT_ACTADR"] = np.array(actadr, i32)
        A["ACT_CTRLLIMITED"] = np.array([int(a.ctrllimited) for a in self.actuators], i32)
        A["ACT_FORCELIMITED"] = np.array([int(a.forcelimited) for a in self.actuators], i32)
        A["ACT_GEAR"] = np.array([a.gear for a in self.actuators], f32)
        A["ACT_DYNPRM"] = np.array([a.dynprm for a in self.actuators], f32).reshape(nu, 3)
        A["ACT_GAINPRM"] = np.array([a.gainprm for a in self.actuators], f32).reshape(nu, 9)
        A["ACT_BIASPRM"] = np.array([a.biasprm for a in self.actuators], f32).reshape(nu, 9)
        A["ACT_CTRLRANGE"] = np.array([a.ctrlrange for a in self.actuators], f32).reshape(nu, 2)
        A["ACT_FORCERANGE"] = np.array([a.forcerange for a in self.actuators], f32).reshape(nu, 2)

        neq = len(self.equalities)
        A["EQ_TYPE"] = np.full(neq, C["MM_EQ_JOINT"], i32)
        A["EQ_OBJ1ID"] = np.array([self._jname[e["j1"]] for e in self.equalities], i32)
        A["EQ_OBJ2ID"] = np.array([self._jname[e["j2"]] if e["j2"] is not None else -1
                                   for e in self.equalities], i32)
        A["EQ_DATA"] = np.array([e["data"] for e in self.equalities], f32).reshape(neq, 5)
        A["EQ_SOLREF"] = np.array([e["solref"] for e in self.equalities], f32).reshape(neq, 2)
        A["EQ_SOLIMP"] = np.array([e["solimp"] for e in self.equalities], f32).reshape(neq, 5)

        for p in self.pairs:   # MuJoCo orders a pair so that type(geom1) <= type(geom2)
            if self.geoms[self._gname[p["g1"]]]["type"] > self.geoms[self._gname[p["g2"]]]["type"]:
                p["g1"], p["g2"] = p["g2"], p["g1"]
        # Entries of the PAIR_* sections: a pair whose collider can return up to FOUR contacts (plane-box, plane-cylinder) takes two
        # consecutive, identical entries -- entry k of the run keeps contacts 2k, 2k+1 -- so that every entry (one lane of the HIP
        # kernel) yields at most two (oracle/mmo_collision.inc)
        def _ptypes(p):
            return self.geoms[self._gname[p["g1"]]]["type"], self.geoms[self._gname[p["g2"]]]["type"]
        entries = []
        for p in self.pairs:
            entries.append(p)
            if _ptypes(p) in ((C["MM_GEOM_PLANE"], C["MM_GEOM_BOX"]), (C["MM_GEOM_PLANE"], C["MM_GEOM_CYLINDER"])):
                entries.append(p)
        npair = len(entries)
        A["PAIR_GEOM1"] = np.array([self._gname[p["g1"]] for p in entries], i32)
        A["PAIR_GEOM2"] = np.array([self._gname[p["g2"]] for p in entries], i32)
        A["PAIR_CONDIM"] = np.array([p["condim"] for p in entries], i32)
        A["PAIR_FRICTION"] = np.array([p["friction"] for p in entries], f32).reshape(npair, 3)
        A["PAIR_MARGIN"] = np.array([p["margin"] for p in entries], f32)
        A["PAIR_GAP"] = np.array([p["gap"] for p in entries], f32)
        A["PAIR_SOLREF"] = np.array([p["solref"] for p in entries], f32).reshape(npair, 2)
        A["PAIR_SOLIMP"] = np.array([p["solimp"] for p in entries], f32).reshape(npair, 5)

        # body levels (depth-sorted) for the cooperative engine
        depth = np.zeros(nbody, i32)
        for i in range(1, nbody):
            depth[i] = depth[parent[i]] + 1
        nlevel = int(depth.max()) if nbody > 1 else 0
        lv_adr = [0]; lv_body = []
        for l in range(1, nlevel + 1):
            lv_body.extend([i for i in range(1, nbody) if depth[i] == l])
            lv_adr.append(len(lv_body))
        A["LEVEL_ADR"] = np.array(lv_adr, i32); A["LEVEL_BODY"] = np.array(lv_body, i32)
        ndl = int(dof_depth.max()) + 1 if nv else 0
        dl_adr = [0]; dl_dof = []
        for l in range(ndl):
            dl_dof.extend([d for d in range(nv) if dof_depth[d] == l])
            dl_adr.append(len(dl_dof))
        A["DOF_LEVEL_ADR"] = np.array(dl_adr, i32); A["DOF_LEVEL_DOF"] = np.array(dl_dof, i32)

        # ---- static sparsity pattern of the tendon Jacobian --------------------
        def body_dofs_chain(b):
            out = []
            while b > 0:
                if body_dofnum[b] > 0:
                    out.extend(range(body_dofadr[b], body_dofadr[b] + body_dofnum[b]))
                b = parent[b]
            return set(out)
        tenj_adr = [0]; tenj_dof = []
        site_body = A["SITE_BODYID"]; geom_body = A["GEOM_BODYID"]
        for ti, t in enumerate(self.tendons):
            dofs = set()
            a0 = tadr[ti]; n = tnum[ti]
            # bodies of consecutive path points (sites and wrap geoms), pulleys break the chain
            seq = []
            for k in range(a0, a0 + n):
                if wt[k] == C["MM_WRAP_SITE"]:
                    seq.append(int(site_body[wo[k]]))
                elif wt[k] in (C["MM_WRAP_SPHERE"], C["MM_WRAP_CYLINDER"]):
                    seq.append(int(geom_body[wo[k]]))
                elif wt[k] == C["MM_WRAP_PULLEY"]:
                    seq.append(None)
                elif wt[k] == C["MM_WRAP_JOINT"]:
                    dofs.add(int(dofadr[wo[k]]))
            for x in range(len(seq) - 1):
                b0, b1 = seq[x], seq[x + 1]
                if b0 is None or b1 is None or b0 == b1:
                    continue
                dofs |= body_dofs_chain(b0) ^ body_dofs_chain(b1)
            # a wrap geom may be skipped (no wrap): then site(k) connects to site(k+2)
            for x in range(len(seq) - 2):
                if seq[x] is not None and seq[x + 2] is not None and seq[x + 1] is not None:
                    k = a0 + x + 1
                    if wt[k] in (C["MM_WRAP_SPHERE"], C["MM_WRAP_CYLINDER"]) and seq[x] != seq[x + 2]:
                        dofs |= body_dofs_chain(seq[x]) ^ body_dofs_chain(seq[x + 2])
            tenj_dof.extend(sorted(dofs)); tenj_adr.append(len(tenj_dof))
        A["TENJ_ADR"] = np.array(tenj_adr, i32); A["TENJ_DOF"] = np.array(tenj_dof, i32)
        ntenJ = len(tenj_dof)

        # ---- static bound on constraint rows --------------------------------------
        nlim_j = int(sum(1 for j in self.joints if j.limited))
        nlim_t = int(sum(1 for t in self.tendons if t.limited))
        con_rows = 0
        for p in self.pairs:
            con_rows = max(con_rows, 1 if p["condim"] == 1 else 2 * (p["condim"] - 1))
        # contacts an entry can produce: two for plane-capsule (one per end cap), capsule-capsule (parallel axes), capsule-box (the
        # second sphere of mjc_CapsuleBox along a face) and each of the two entries of a plane-box / plane-cylinder pair; one otherwise
        two = ((C["MM_GEOM_PLANE"], C["MM_GEOM_CAPSULE"]), (C["MM_GEOM_CAPSULE"], C["MM_GEOM_CAPSULE"]),
               (C["MM_GEOM_PLANE"], C["MM_GEOM_BOX"]), (C["MM_GEOM_PLANE"], C["MM_GEOM_CYLINDER"]),
               (C["MM_GEOM_CAPSULE"], C["MM_GEOM_BOX"]), (C["MM_GEOM_BOX"], C["MM_GEOM_CAPSULE"]))
        ncon_bound = sum(2 if _ptypes(p) in two else 1 for p in entries)
        nconmax = self.nconmax if self.nconmax else ncon_bound
        nfric = int(np.count_nonzero(dof_floss > 0))
        njmax = neq + nfric + nlim_j + nlim_t + nconmax * con_rows
        if self.njmax:      # explicit row bound (mjModel.njmax is independent of nconmax): rows beyond it are dropped and flagged
            njmax = int(self.njmax)

        oi = np.zeros(C["MM_OI_COUNT"], i32)
        oi[C["MM_OI_NQ"]] = nq; oi[C["MM_OI_NV"]] = nv; oi[C["MM_OI_NU"]] = nu; oi[C["MM_OI_NA"]] = na
        oi[C["MM_OI_NBODY"]] = nbody; oi[C["MM_OI_NJNT"]] = njnt; oi[C["MM_OI_NGEOM"]] = ngeom
        oi[C["MM_OI_NSITE"]] = nsite; oi[C["MM_OI_NTENDON"]] = ntendon; oi[C["MM_OI_NWRAP"]] = nwrap
        oi[C["MM_OI_NEQ"]] = neq; oi[C["MM_OI_NPAIR"]] = npair; oi[C["MM_OI_NM"]] = nM
        oi[C["MM_OI_NLEVEL"]] = nlevel; oi[C["MM_OI_ITERATIONS"]] = self.iterations
        oi[C["MM_OI_LS_ITERATIONS"]] = self.ls_iterations; oi[C["MM_OI_INTEGRATOR"]] = self.integrator
        oi[C["MM_OI_EULERDAMP"]] = int(self.eulerdamp); oi[C["MM_OI_NJMAX"]] = njmax
        oi[C["MM_OI_NTENJ"]] = ntenJ; oi[C["MM_OI_NCONMAX"]] = nconmax
        of = np.zeros(C["MM_OF_COUNT"], f32)
        of[C["MM_OF_TIMESTEP"]] = self.timestep
        of[C["MM_OF_GRAV_X"]:C["MM_OF_GRAV_Z"] + 1] = self.gravity
        of[C["MM_OF_TOLERANCE"]] = self.tolerance; of[C["MM_OF_LS_TOLERANCE"]] = self.ls_tolerance
        of[C["MM_OF_IMPRATIO"]] = 1.0
        A["OPT_I"] = oi; A["OPT_F"] = of

        # ---- derived constants at qpos0 (MuJoCo compiler's set0 stage) -----------
        km = K.KinModel(A, nq, nv, nbody)
        q0 = qpos0.astype(np.float64)[None, :]
        M0 = km.mass_matrix(q0)[0]
        M0 = M0 + np.diag(dof_armature.astype(np.float64))
        Minv = np.linalg.inv(M0) if nv else np.zeros((0, 0))
        of[C["MM_OF_MEANINERTIA"]] = float(np.trace(M0) / max(1, nv)) if nv else 1.0
        dof_inv = np.zeros(nv, f32)
        for ji, j in enumerate(self.joints):
            d0 = dofadr[ji]
            if j.type == C["MM_JNT_FREE"]:
                dof_inv[d0:d0 + 3] = np.mean(np.diag(Minv)[d0:d0 + 3])
                dof_inv[d0 + 3:d0 + 6] = np.mean(np.diag(Minv)[d0 + 3:d0 + 6])
            elif j.type == C["MM_JNT_BALL"]:
                dof_inv[d0:d0 + 3] = np.mean(np.diag(Minv)[d0:d0 + 3])
            else:
                dof_inv[d0] = Minv[d0, d0]
        A["DOF_INVWEIGHT0"] = dof_inv
        body_inv = np.zeros((nbody, 2), f32)
        Jp, Jr = km.body_com_jacobians(q0)
        for b in range(1, nbody):
            if len(body_dofs_chain(b)) == 0:
                continue
            Ap = Jp[0, b] @ Minv @ Jp[0, b].T
            Ar = Jr[0, b] @ Minv @ Jr[0, b].T
            body_inv[b, 0] = max(C_MINVAL, np.trace(Ap) / 3.0)
            body_inv[b, 1] = max(C_MINVAL, np.trace(Ar) / 3.0)
        A["BODY_INVWEIGHT0"] = body_inv

        # tendon Jacobian at qpos0 by central differences of the (wrap-aware) length
        ten_inv = np.zeros(ntendon, f32)
        J0 = np.zeros((ntendon, nv))
        if ntendon and nv:
            J0 = km.tendon_jacobian_fd(q0)[0]
            for t in range(ntendon):
                ten_inv[t] = max(C_MINVAL, float(J0[t] @ Minv @ J0[t]))
        A["TENDON_INVWEIGHT0"] = ten_inv
        L0 = km.tendon_length(q0)[0] if ntendon else np.zeros(0)
        ls = np.array([t.springlength for t in self.tendons], np.float64).reshape(ntendon, 2)
        for t in range(ntendon):  # MuJoCo: springlength -1 means "use length at qpos0"
            if ls[t, 0] < 0:
                ls[t] = (L0[t], L0[t])
        A["TENDON_LENGTHSPRING"] = ls.astype(f32)

        # muscle length ranges: extreme tendon lengths over the joint-limit box
        lr = np.zeros((nu, 2), f32); acc0 = np.zeros(nu, f32)
        need_lr = [i for i, a in enumerate(self.actuators)
                   if a.trntype == C["MM_TRN_TENDON"] and a.lengthrange is None]
        if need_lr:
            rng = np.random.default_rng(seed)
            lo = np.zeros(nq); hi = np.zeros(nq)
            for ji, j in enumerate(self.joints):
                if j.type in (C["MM_JNT_SLIDE"], C["MM_JNT_HINGE"]):
                    if j.limited:
                        lo[qposadr[ji]], hi[qposadr[ji]] = j.range
                    elif j.type == C["MM_JNT_HINGE"] and not any(e["j1"] == j.name for e in self.equalities):
                        lo[qposadr[ji]], hi[qposadr[ji]] = (-np.pi, np.pi)
                    else:   # unlimited slide, or a coordinate driven by a joint equality: reference value
                        lo[qposadr[ji]] = hi[qposadr[ji]] = qpos0[qposadr[ji]]
            free_mask = np.ones(nq, bool)
            qs = lo + (hi - lo) * rng.random((lengthrange_samples, nq))
            # corners-ish: each coordinate snapped to an end with prob 1/2 in half of the samples
            snap = rng.random((lengthrange_samples, nq)) < 0.5
            ends = np.where(rng.random((lengthrange_samples, nq)) < 0.5, lo, hi)
            half = lengthrange_samples // 2
            qs[:half] = np.where(snap[:half], ends[:half], qs[:half])
            for ji, j in enumerate(self.joints):  # keep quaternion coordinates at reference
                if j.type in (C["MM_JNT_FREE"], C["MM_JNT_BALL"]):
                    n = 7 if j.type == C["MM_JNT_FREE"] else 4
                    qs[:, qposadr[ji]:qposadr[ji] + n] = qpos0[qposadr[ji]:qposadr[ji] + n]
            for e in self.equalities:   # coupled coordinates follow their polynomial
                a1 = qposadr[self._jname[e["j1"]]]
                if e["j2"] is None:
                    qs[:, a1] = qpos0[a1] + e["data"][0]
                else:
                    a2 = qposadr[self._jname[e["j2"]]]
                    x = qs[:, a2] - qpos0[a2]
                    qs[:, a1] = qpos0[a1] + sum(c * x ** i for i, c in enumerate(e["data"]))
            Ls = km.tendon_length(qs)
            for i in need_lr:
                t = trnid[i]; g = self.actuators[i].gear
                a, b = g * Ls[:, t].min(), g * Ls[:, t].max()
                lr[i] = (min(a, b), max(a, b))
        for i, a in enumerate(self.actuators):
            if a.lengthrange is not None:
                lr[i] = a.lengthrange
            if a.trntype == C["MM_TRN_TENDON"]:
                mom = a.gear * J0[trnid[i]]
            else:
                mom = np.zeros(nv); mom[dofadr[trnid[i]]] = a.gear
            acc0[i] = float(np.linalg.norm(Minv @ mom)) if nv else 0.0
        A["ACT_LENGTHRANGE"] = lr; A["ACT_ACC0"] = acc0

        names = dict(body=dict(self._bname), joint=dict(self._jname), site=dict(self._sname),
                     geom=dict(self._gname), tendon=dict(self._tname),
                     actuator={a.name: i for i, a in enumerate(self.actuators)})
        return CompiledModel(self.name, A, names)


C_MINVAL = 1e-15
