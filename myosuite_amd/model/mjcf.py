"""MJCF subset importer / exporter for the engine's ModelSpec  (SURVEY.md 8f row 2).

``load(path)`` parses the part of MuJoCo's XML format that MyoSuite's musculoskeletal models use and the engine
implements -- compiler (angle, eulerseq, autolimits, inertiafromgeom for primitives), option (timestep, gravity, integrator
Euler / RK4, iterations, tolerance), nested ``<default>`` classes with ``childclass``, ``<include>``, the body tree with
``<inertial>`` / ``<joint>`` / ``<freejoint>`` / ``<geom>`` / ``<site>``, spatial tendons (site / geom+sidesite / pulley) and
fixed tendons, ``<general>`` / ``<muscle>`` / ``<motor>`` actuators, joint equalities, explicit contact pairs plus pairs
generated from contype / conaffinity (with ``<exclude>``), keyframes -- into a ``ModelSpec``; ``dump(spec)`` writes a spec back
as MJCF (radians, explicit inertials, explicit contact pairs).  Elements that do not affect the physics path (asset, visual,
camera, light, sensor, rgba, material, group ...) are ignored; physics features the engine does not implement raise
``MjcfError`` (never silently dropped): mesh / hfield collision geoms, ball joints with limits, weld / connect equalities,
elliptic cones, the full `implicit` integrator (`implicitfast` is implemented).

Written from MuJoCo's public XML reference; it does not use or need the ``mujoco`` package.  The real ``myo_sim`` MJCF is an
empty submodule in the reference checkout, so the tests exercise the importer on MJCF written by ``dump`` (round trip of the
synthetic models) and on hand-written snippets that cover defaults / includes / degrees / fromto / inertiafromgeom.
``tools/validate_against_mujoco.py`` is the other half of that row: where ``mujoco`` is installed it loads the same XML in
libmujoco and compares trajectories with the oracle and the HIP engine.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

from .spec import C, DEFAULT_SOLIMP, DEFAULT_SOLREF, ModelSpec, _Actuator


class MjcfError(ValueError):
    pass


# ----------------------------------------------------------------------------- small helpers
def _floats(s: Optional[str], n: Optional[int] = None, default=None):
    if s is None:
        return default
    v = [float(x) for x in s.replace(",", " ").split()]
    if n is not None and len(v) != n:
        if len(v) < n and default is not None:       # MuJoCo pads short vectors with the default's tail
            v = v + list(default)[len(v):]
        else:
            raise MjcfError(f"expected {n} numbers, got {s!r}")
    return v


def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _mat2quat(R):
    R = np.asarray(R, float)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def _z_to(v):
    v = np.asarray(v, float) / np.linalg.norm(v)
    ax = np.cross([0.0, 0.0, 1.0], v)
    sn, cs = np.linalg.norm(ax), v[2]
    if sn < 1e-12:
        return np.array([1.0, 0, 0, 0]) if cs > 0 else np.array([0.0, 1.0, 0, 0])
    ang = math.atan2(sn, cs)
    return np.concatenate([[math.cos(ang / 2)], math.sin(ang / 2) * ax / sn])


class _Ctx:
    """compiler settings + default classes"""

    def __init__(self):
        self.degree = True
        self.eulerseq = "xyz"
        self.autolimits = True
        self.inertiafromgeom = "auto"
        self.defaults: Dict[str, Dict[str, Dict[str, str]]] = {"main": {}}
        self.parent_class: Dict[str, str] = {}
        # lenient: the file is known to be incomplete (includes skipped: `missing_include="skip"`, dry_run) -- a default class that
        # only a missing include defines resolves to "main", and what could not be interpreted is counted in `skipped`
        self.lenient = False
        self.skipped: Dict[str, int] = {}

    def ang(self, x):
        return math.radians(x) if self.degree else x

    def orientation(self, a: Dict[str, str]) -> np.ndarray:
        if "quat" in a:
            q = np.array(_floats(a["quat"], 4))
            return q / np.linalg.norm(q)
        if "euler" in a:
            e = [self.ang(x) for x in _floats(a["euler"], 3)]
            q = np.array([1.0, 0, 0, 0])
            for ch, ang in zip(self.eulerseq, e):
                axis = {"x": 0, "y": 1, "z": 2}[ch.lower()]
                r = np.zeros(4); r[0] = math.cos(ang / 2); r[1 + axis] = math.sin(ang / 2)
                q = _qmul(q, r) if ch.islower() else _qmul(r, q)     # lower case: intrinsic (rotating frame)
            return q
        if "axisangle" in a:
            v = _floats(a["axisangle"], 4)
            ax = np.array(v[:3]) / np.linalg.norm(v[:3]); ang = self.ang(v[3])
            return np.concatenate([[math.cos(ang / 2)], math.sin(ang / 2) * ax])
        if "xyaxes" in a:
            v = _floats(a["xyaxes"], 6)
            x = np.array(v[:3]); x /= np.linalg.norm(x)
            y = np.array(v[3:]); y -= x * (x @ y); y /= np.linalg.norm(y)
            return _mat2quat(np.stack([x, y, np.cross(x, y)], 1))
        if "zaxis" in a:
            return _z_to(_floats(a["zaxis"], 3))
        return np.array([1.0, 0, 0, 0])

    # ---- defaults
    def read_defaults(self, node: ET.Element, cls: str = "main", parent: Optional[str] = None):
        if cls not in self.defaults:
            self.defaults[cls] = {}
        if parent is not None:
            self.parent_class[cls] = parent
        for ch in node:
            if ch.tag == "default":
                sub = ch.attrib.get("class")
                if sub is None:
                    raise MjcfError("nested <default> needs a class")
                self.read_defaults(ch, sub, cls)
            else:
                self.defaults[cls].setdefault(ch.tag, {}).update(ch.attrib)

    def resolve(self, tag: str, attrib: Dict[str, str], childclass: Optional[str]) -> Dict[str, str]:
        cls = attrib.get("class", childclass or "main")
        if cls not in self.defaults:
            if not self.lenient:
                raise MjcfError(f"unknown default class {cls!r}")
            self.skipped[f"default class {cls!r} (defined by a missing include)"] = self.skipped.get(f"default class {cls!r} (defined by a missing include)", 0) + 1
            cls = "main"
        chain = []
        c = cls
        while c is not None:
            chain.append(c)
            c = self.parent_class.get(c)
        out: Dict[str, str] = {}
        for c in reversed(chain):
            out.update(self.defaults[c].get(tag, {}))
        out.update(attrib)
        return out


# ----------------------------------------------------------------------------- inertia of primitive geoms
def _geom_inertia(gtype: str, size, mass: Optional[float], density: float):
    """(mass, principal inertia about the geom frame axes) of a solid primitive"""
    if gtype == "sphere":
        r = size[0]; vol = 4.0 / 3.0 * math.pi * r ** 3
        m = mass if mass is not None else density * vol
        i = 0.4 * m * r * r
        return m, np.array([i, i, i])
    if gtype == "capsule":
        r, h = size[0], size[1]
        vc, vs = math.pi * r * r * 2 * h, 4.0 / 3.0 * math.pi * r ** 3
        m = mass if mass is not None else density * (vc + vs)
        mc, ms = m * vc / (vc + vs), m * vs / (vc + vs)
        iz = 0.5 * mc * r * r + 0.4 * ms * r * r
        # two hemispheres about the capsule centre: 2/5 ms r^2 + ms (h^2 + 3/4 r h)  (parallel axis, centroid at 3r/8)
        ix = mc * (3 * r * r + 4 * h * h) / 12.0 + ms * (0.4 * r * r + h * h + 0.75 * r * h)
        return m, np.array([ix, ix, iz])
    if gtype == "cylinder":
        r, h = size[0], size[1]
        m = mass if mass is not None else density * math.pi * r * r * 2 * h
        return m, np.array([m * (3 * r * r + 4 * h * h) / 12.0] * 2 + [0.5 * m * r * r])
    if gtype == "box":
        a, b, c = size[:3]
        m = mass if mass is not None else density * 8 * a * b * c
        return m, np.array([m * (b * b + c * c) / 3.0, m * (a * a + c * c) / 3.0, m * (a * a + b * b) / 3.0])
    if gtype == "ellipsoid":
        a, b, c = size[:3]
        m = mass if mass is not None else density * 4.0 / 3.0 * math.pi * a * b * c
        return m, np.array([m * (b * b + c * c) / 5.0, m * (a * a + c * c) / 5.0, m * (a * a + b * b) / 5.0])
    raise MjcfError(f"cannot infer inertia from geom type {gtype!r}")


def _principal(I: np.ndarray):
    w, V = np.linalg.eigh(I)
    order = np.argsort(-w)
    w, V = w[order], V[:, order]
    if np.linalg.det(V) < 0:
        V[:, 2] = -V[:, 2]
    return w, _mat2quat(V)


# ----------------------------------------------------------------------------- loader
def _expand_includes(node: ET.Element, base: str, missing: str, include_map=None):
    i = 0
    while i < len(node):
        ch = node[i]
        if ch.tag == "include":
            fn = ch.attrib["file"]
            for old, new in (include_map or {}).items():     # e.g. the empty simhive/myo_sim submodule -> a local model tree
                if old in fn:
                    fn = os.path.join(new, fn[fn.index(old) + len(old):].lstrip("/"))
                    break
            path = fn if os.path.isabs(fn) else os.path.join(base, fn)
            node.remove(ch)
            if not os.path.exists(path):
                if missing == "skip":
                    continue
                raise MjcfError(f"<include> file not found: {path}")
            sub = ET.parse(path).getroot()
            _expand_includes(sub, os.path.dirname(path), missing, include_map)
            kids = list(sub) if sub.tag in ("mujoco", "mujocoinclude") else [sub]
            for k, el in enumerate(kids):
                node.insert(i + k, el)
            i += len(kids)
        else:
            _expand_includes(ch, base, missing, include_map)
            i += 1


def load(source: str, missing_include: str = "error", include_map: Optional[Dict[str, str]] = None) -> ModelSpec:
    """MJCF file path (or XML string) -> ModelSpec.  ``include_map`` {substring of an <include file=...>: replacement directory}
    resolves includes that point into a tree living elsewhere -- the reference's task XMLs include
    ``../../../../simhive/myo_sim/...`` (an empty git submodule in the reference checkout): pass
    ``{"simhive/myo_sim": "/path/to/myo_sim"}`` (or set MYOSUITE_MYO_SIM_ROOT) to load them against a real or stand-in tree."""
    if include_map is None and os.environ.get("MYOSUITE_MYO_SIM_ROOT"):
        include_map = {"simhive/myo_sim": os.environ["MYOSUITE_MYO_SIM_ROOT"]}
    if os.path.exists(source):
        root = ET.parse(source).getroot(); base = os.path.dirname(os.path.abspath(source))
    else:
        root = ET.fromstring(source); base = os.getcwd()
    if root.tag != "mujoco":
        raise MjcfError("root element must be <mujoco>")
    _expand_includes(root, base, missing_include, include_map)
    ctx = _Ctx()
    ctx.lenient = missing_include == "skip"
    opt = dict(timestep=0.002, gravity=(0.0, 0.0, -9.81), integrator=0, iterations=100, tolerance=1e-8, ls_iterations=50,
               ls_tolerance=0.01, eulerdamp=True)
    for el in root.findall("compiler"):
        a = el.attrib
        if "angle" in a:
            ctx.degree = a["angle"] == "degree"
        if "eulerseq" in a:
            ctx.eulerseq = a["eulerseq"]
        if "autolimits" in a:
            ctx.autolimits = a["autolimits"] == "true"
        if "inertiafromgeom" in a:
            ctx.inertiafromgeom = a["inertiafromgeom"]
        if a.get("coordinate", "local") != "local":
            raise MjcfError("only local coordinates are supported")
    for el in root.findall("option"):
        a = el.attrib
        if "timestep" in a:
            opt["timestep"] = float(a["timestep"])
        if "gravity" in a:
            opt["gravity"] = tuple(_floats(a["gravity"], 3))
        if "integrator" in a:
            if a["integrator"] not in ("Euler", "RK4", "implicitfast"):
                raise MjcfError(f"integrator {a['integrator']!r} is not implemented (Euler, RK4, implicitfast)")
            opt["integrator"] = {"Euler": 0, "RK4": 1, "implicitfast": 3}[a["integrator"]]
        if a.get("cone", "pyramidal") != "pyramidal":
            raise MjcfError("only pyramidal friction cones are implemented")
        if a.get("solver", "Newton") != "Newton":
            raise MjcfError("only the Newton solver is implemented")
        for k in ("iterations", "ls_iterations"):
            if k in a:
                opt[k] = int(a[k])
        for k in ("tolerance", "ls_tolerance"):
            if k in a:
                opt[k] = float(a[k])
        for fl in el.findall("flag"):
            if fl.attrib.get("eulerdamp") == "disable":
                opt["eulerdamp"] = False
    for el in root.findall("default"):
        ctx.read_defaults(el)
    nconmax = njmax = 0
    for el in root.findall("size"):
        if int(el.attrib.get("nconmax", "-1")) > 0:
            nconmax = int(el.attrib["nconmax"])
        if int(el.attrib.get("njmax", "-1")) > 0:          # the row bound is independent of the contact bound (mjModel.njmax)
            njmax = int(el.attrib["njmax"])

    s = ModelSpec(root.attrib.get("model", "mjcf"), timestep=opt["timestep"], gravity=opt["gravity"],
                  tolerance=opt["tolerance"], iterations=opt["iterations"], ls_iterations=opt["ls_iterations"],
                  ls_tolerance=opt["ls_tolerance"], integrator=opt["integrator"], eulerdamp=opt["eulerdamp"], nconmax=nconmax)
    if njmax:
        s.njmax = njmax
    geom_info: List[dict] = []       # contact attributes of every geom (for the generated pairs)
    auto = [0]

    def uname(prefix):
        auto[0] += 1
        return f"_{prefix}{auto[0]}"

    def add_geom(el, body, childclass):
        a = ctx.resolve("geom", el.attrib, childclass)
        gtype = a.get("type", "sphere")
        if gtype in ("mesh", "hfield", "sdf"):
            if int(a.get("contype", "1")) or int(a.get("conaffinity", "1")):
                raise MjcfError(f"collision geom of type {gtype!r} is not implemented (geom {a.get('name')})")
            return None, None
        size = _floats(a.get("size"), None, [0.0]) or [0.0]
        pos = np.array(_floats(a.get("pos"), 3, [0.0, 0, 0]))
        quat = ctx.orientation(a)
        if "fromto" in a:
            ft = np.array(_floats(a["fromto"], 6))
            p0, p1 = ft[:3], ft[3:]
            pos = 0.5 * (p0 + p1); quat = _z_to(p1 - p0)
            size = [size[0], 0.5 * float(np.linalg.norm(p1 - p0))] + list(size[2:])
        name = a.get("name") or uname("geom")
        s.add_geom(name, body, gtype, size[:3], pos=tuple(pos), quat=tuple(quat))
        info = dict(name=name, body=body, type=gtype, contype=int(a.get("contype", "1")),
                    conaffinity=int(a.get("conaffinity", "1")), condim=int(a.get("condim", "3")),
                    friction=_floats(a.get("friction"), 3, [1.0, 0.005, 0.0001]), margin=float(a.get("margin", "0")),
                    gap=float(a.get("gap", "0")), solref=_floats(a.get("solref"), 2, list(DEFAULT_SOLREF)),
                    solimp=_floats(a.get("solimp"), 5, list(DEFAULT_SOLIMP)), solmix=float(a.get("solmix", "1")),
                    priority=int(a.get("priority", "0")))
        geom_info.append(info)
        mass = float(a["mass"]) if "mass" in a else None
        return dict(type=gtype, size=size, pos=pos, quat=quat, mass=mass, density=float(a.get("density", "1000"))), info

    def add_body(el, parent, childclass):
        a = el.attrib
        cc = a.get("childclass", childclass)
        name = a.get("name") or uname("body")
        pos = tuple(_floats(a.get("pos"), 3, [0.0, 0, 0]))
        quat = ctx.orientation(a)
        inert = el.find("inertial")
        geoms_for_inertia = []
        # inertial first (add_body needs it); geoms that may define it are parsed twice: cheap
        if inert is not None and ctx.inertiafromgeom != "true":
            ia = inert.attrib
            mass = float(ia["mass"])
            ipos = tuple(_floats(ia.get("pos"), 3, [0.0, 0, 0]))
            iq = ctx.orientation(ia)
            if "fullinertia" in ia:
                f = _floats(ia["fullinertia"], 6)
                I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                diag, q2 = _principal(I)
                iq = _qmul(iq, q2)
            else:
                diag = np.array(_floats(ia["diaginertia"], 3))
        else:
            mass, ipos, iq, diag = 0.0, (0.0, 0, 0), np.array([1.0, 0, 0, 0]), np.zeros(3)
            if ctx.inertiafromgeom != "false":
                for g in el.findall("geom"):
                    ga = ctx.resolve("geom", g.attrib, cc)
                    gt = ga.get("type", "sphere")
                    if gt in ("mesh", "hfield", "sdf", "plane"):
                        if gt == "mesh":
                            raise MjcfError(f"body {name!r} has no <inertial> and a mesh geom: inertia from meshes is not implemented")
                        continue
                    size = _floats(ga.get("size"), None, [0.0])
                    gpos = np.array(_floats(ga.get("pos"), 3, [0.0, 0, 0])); gq = ctx.orientation(ga)
                    if "fromto" in ga:
                        ft = np.array(_floats(ga["fromto"], 6))
                        gpos = 0.5 * (ft[:3] + ft[3:]); gq = _z_to(ft[3:] - ft[:3])
                        size = [size[0], 0.5 * float(np.linalg.norm(ft[3:] - ft[:3]))]
                    m, Ig = _geom_inertia(gt, size, float(ga["mass"]) if "mass" in ga else None, float(ga.get("density", "1000")))
                    geoms_for_inertia.append((m, gpos, _quat2mat(gq) @ np.diag(Ig) @ _quat2mat(gq).T))
                if geoms_for_inertia:
                    mass = sum(m for m, _, _ in geoms_for_inertia)
                    com = sum(m * p for m, p, _ in geoms_for_inertia) / mass
                    I = np.zeros((3, 3))
                    for m, p, Ig in geoms_for_inertia:
                        r = p - com
                        I += Ig + m * ((r @ r) * np.eye(3) - np.outer(r, r))
                    diag, iq = _principal(I)
                    ipos = tuple(com)
        s.add_body(name, parent, pos=pos, quat=tuple(quat), mass=mass, ipos=ipos, inertia=tuple(diag), iquat=tuple(iq))
        for j in list(el.findall("freejoint")) + list(el.findall("joint")):
            if j.tag == "freejoint":
                s.add_joint(j.attrib.get("name") or uname("jnt"), name, "free")
                continue
            ja = ctx.resolve("joint", j.attrib, cc)
            jt = ja.get("type", "hinge")
            rng = _floats(ja.get("range"), 2)
            limited = ja.get("limited", "auto")
            lim = (limited == "true") or (limited == "auto" and ctx.autolimits and rng is not None)
            angular = jt in ("hinge", "ball")
            if lim and jt == "ball":
                raise MjcfError("limited ball joints are not implemented")
            if rng is not None and angular:
                rng = [ctx.ang(x) for x in rng]
            kw = dict(pos=tuple(_floats(ja.get("pos"), 3, [0.0, 0, 0])), axis=tuple(_floats(ja.get("axis"), 3, [0.0, 0, 1])),
                      range=tuple(rng) if lim else None, stiffness=float(ja.get("stiffness", "0")),
                      damping=float(ja.get("damping", "0")), armature=float(ja.get("armature", "0")),
                      margin=float(ja.get("margin", "0")),
                      ref=ctx.ang(float(ja.get("ref", "0"))) if angular else float(ja.get("ref", "0")),
                      springref=ctx.ang(float(ja.get("springref", "0"))) if angular else float(ja.get("springref", "0")),
                      solref=tuple(_floats(ja.get("solreflimit"), 2, list(DEFAULT_SOLREF))),
                      solimp=tuple(_floats(ja.get("solimplimit"), 5, list(DEFAULT_SOLIMP))),
                      frictionloss=float(ja.get("frictionloss", "0")),
                      solreffriction=tuple(_floats(ja.get("solreffriction"), 2, list(DEFAULT_SOLREF))),
                      solimpfriction=tuple(_floats(ja.get("solimpfriction"), 5, list(DEFAULT_SOLIMP))))
            s.add_joint(ja.get("name") or uname("jnt"), name, jt, **kw)
        for g in el.findall("geom"):
            add_geom(g, name, cc)
        for st in el.findall("site"):
            sa = ctx.resolve("site", st.attrib, cc)
            s.add_site(sa.get("name") or uname("site"), name, tuple(_floats(sa.get("pos"), 3, [0.0, 0, 0])))
        for b in el.findall("body"):
            add_body(b, name, cc)

    for wb in root.findall("worldbody"):
        for g in wb.findall("geom"):
            add_geom(g, "world", None)
        for st in wb.findall("site"):
            sa = ctx.resolve("site", st.attrib, None)
            s.add_site(sa.get("name") or uname("site"), "world", tuple(_floats(sa.get("pos"), 3, [0.0, 0, 0])))
        for b in wb.findall("body"):
            add_body(b, "world", None)

    # ---- tendons
    for tn in root.findall("tendon"):
        for el in tn:
            a = ctx.resolve("tendon", el.attrib, None)
            name = a.get("name") or uname("tendon")
            rng = _floats(a.get("range"), 2)
            limited = a.get("limited", "auto")
            tlim = limited == "true" or (limited == "auto" and ctx.autolimits and rng is not None)
            path = []
            if el.tag == "spatial":
                for p in el:
                    if p.tag == "site":
                        path.append(("site", p.attrib["site"]))
                    elif p.tag == "geom":
                        gname = p.attrib["geom"]
                        gt = s.geoms[s._gname[gname]]["type"]
                        path.append(("sphere" if gt == 2 else "cylinder", gname, p.attrib.get("sidesite")))
                    elif p.tag == "pulley":
                        path.append(("pulley", float(p.attrib["divisor"])))
            elif el.tag == "fixed":
                for p in el.findall("joint"):
                    path.append(("joint", p.attrib["joint"], float(p.attrib["coef"])))
            else:
                continue
            sl = _floats(a.get("springlength"), None)
            sl = (-1.0, -1.0) if sl is None else ((sl[0], sl[0]) if len(sl) == 1 else (sl[0], sl[1]))
            s.add_tendon(name, path, stiffness=float(a.get("stiffness", "0")), damping=float(a.get("damping", "0")),
                         springlength=sl, limited=tlim, range=tuple(rng) if tlim else (0.0, 0.0),
                         margin=float(a.get("margin", "0")),
                         solref=tuple(_floats(a.get("solreflimit"), 2, list(DEFAULT_SOLREF))),
                         solimp=tuple(_floats(a.get("solimplimit"), 5, list(DEFAULT_SOLIMP))))

    # ---- actuators
    for ac in root.findall("actuator"):
        for el in ac:
            a = ctx.resolve(el.tag, el.attrib, None)
            if el.tag != "general":                    # shortcuts also inherit <general> defaults
                g = ctx.resolve("general", {k: v for k, v in el.attrib.items() if k == "class"}, None)
                g.update(a); a = g
            name = a.get("name") or uname("act")
            if "tendon" in a:
                trn, target = C["MM_TRN_TENDON"], a["tendon"]
            elif "joint" in a:
                trn, target = C["MM_TRN_JOINT"], a["joint"]
            else:
                raise MjcfError(f"actuator {name}: only joint and tendon transmissions are implemented")
            gear = _floats(a.get("gear"), None, [1.0])[0]
            cr = _floats(a.get("ctrlrange"), 2)
            cl = a.get("ctrllimited", "auto")
            ctrllimited = cl == "true" or (cl == "auto" and ctx.autolimits and cr is not None)
            fr = _floats(a.get("forcerange"), 2)
            fl = a.get("forcelimited", "auto")
            forcelimited = fl == "true" or (fl == "auto" and ctx.autolimits and fr is not None)
            lr = _floats(a.get("lengthrange"), 2)
            z9 = [0.0] * 9
            if el.tag == "muscle" or a.get("dyntype") == "muscle" or a.get("gaintype") == "muscle":
                if el.tag == "muscle":
                    tc = _floats(a.get("timeconst"), 2, [0.01, 0.04])
                    dyn = (tc[0], tc[1], float(a.get("tausmooth", "0")))
                    r = _floats(a.get("range"), 2, [0.75, 1.05])
                    prm = [r[0], r[1], float(a.get("force", "-1")), float(a.get("scale", "200")), float(a.get("lmin", "0.5")),
                           float(a.get("lmax", "1.6")), float(a.get("vmax", "1.5")), float(a.get("fpmax", "1.3")),
                           float(a.get("fvmax", "1.2"))]
                    gain = bias = tuple(prm)
                    if cr is None:
                        cr, ctrllimited = [0.0, 1.0], True
                else:
                    dyn = tuple(_floats(a.get("dynprm"), None, [0.01, 0.04, 0.0])[:3])
                    gain = tuple((_floats(a.get("gainprm"), None, z9) + z9)[:9])
                    bias = tuple((_floats(a.get("biasprm"), None, z9) + z9)[:9])
                s.actuators.append(_Actuator(name, trn, target, float(gear), C["MM_DYN_MUSCLE"], C["MM_GAIN_MUSCLE"],
                                             C["MM_BIAS_MUSCLE"], tuple(dyn), gain, bias, ctrllimited,
                                             tuple(cr) if cr else (0.0, 0.0), forcelimited, tuple(fr) if fr else (0.0, 0.0),
                                             tuple(lr) if lr else None))
            else:
                dyn_name = a.get("dyntype", "none")
                if dyn_name not in ("none", "integrator", "filter"):
                    raise MjcfError(f"actuator {name}: dyntype {dyn_name!r} is not implemented")
                dynt = {"none": C["MM_DYN_NONE"], "integrator": C["MM_DYN_INTEGRATOR"], "filter": C["MM_DYN_FILTER"]}[dyn_name]
                dynp = tuple((_floats(a.get("dynprm"), None, [1.0, 0.0, 0.0]) + [0.0, 0.0])[:3])
                if el.tag == "motor" or (el.tag == "general" and a.get("gaintype", "fixed") == "fixed" and a.get("biastype", "none") == "none"):
                    gp = _floats(a.get("gainprm"), None, [1.0])
                    s.actuators.append(_Actuator(name, trn, target, float(gear), dynt, C["MM_GAIN_FIXED"],
                                                 C["MM_BIAS_NONE"], dynp, tuple(([gp[0]] + z9)[:9]), tuple(z9),
                                                 ctrllimited, tuple(cr) if cr else (0.0, 0.0), forcelimited,
                                                 tuple(fr) if fr else (0.0, 0.0), tuple(lr) if lr else None))
                else:   # position / velocity shortcuts and <general biastype="affine">: fixed gain + affine bias
                    if a.get("gaintype", "fixed") != "fixed" or a.get("biastype", "affine" if el.tag != "general" else "none") not in ("affine", "none"):
                        raise MjcfError(f"actuator {name}: gaintype / biastype combination is not implemented")
                    if el.tag == "position":
                        kp = float(a.get("kp", "1")); kv = float(a.get("kv", "0"))
                        gp, bp = [kp], [0.0, -kp, -kv]
                    elif el.tag == "velocity":
                        kv = float(a.get("kv", "1"))
                        gp, bp = [kv], [0.0, 0.0, -kv]
                    else:
                        gp = _floats(a.get("gainprm"), None, [1.0]); bp = (_floats(a.get("biasprm"), None, [0.0]) + [0.0] * 3)[:3]
                    s.actuators.append(_Actuator(name, trn, target, float(gear), dynt, C["MM_GAIN_FIXED"],
                                                 C["MM_BIAS_AFFINE"], dynp, tuple(([gp[0]] + z9)[:9]),
                                                 tuple((list(bp) + z9)[:9]), ctrllimited, tuple(cr) if cr else (0.0, 0.0),
                                                 forcelimited, tuple(fr) if fr else (0.0, 0.0), tuple(lr) if lr else None))

    # ---- equalities
    for eq in root.findall("equality"):
        for el in eq:
            a = ctx.resolve("equality", el.attrib, None)
            if a.get("active", "true") != "true":
                continue
            if el.tag != "joint":
                raise MjcfError(f"equality <{el.tag}> is not implemented (joint couplings only)")
            pc = _floats(a.get("polycoef"), None, [0.0, 1.0, 0, 0, 0])
            s.add_equality_joint(a["joint1"], a.get("joint2"), pc, solref=tuple(_floats(a.get("solref"), 2, list(DEFAULT_SOLREF))),
                                 solimp=tuple(_floats(a.get("solimp"), 5, list(DEFAULT_SOLIMP))))

    # ---- contacts: explicit pairs, then the pairs MuJoCo's filter would generate
    excl = set()
    explicit = set()
    for ct in root.findall("contact"):
        for el in ct.findall("exclude"):
            excl.add(frozenset((el.attrib["body1"], el.attrib["body2"])))
        for el in ct.findall("pair"):
            a = ctx.resolve("pair", el.attrib, None)
            g1 = next(g for g in geom_info if g["name"] == a["geom1"]); g2 = next(g for g in geom_info if g["name"] == a["geom2"])
            mix = _mix(g1, g2)
            if int(a.get("condim", mix["condim"])) not in (1, 3, 4):
                raise MjcfError(f"<pair {a['geom1']} {a['geom2']}>: condim {a.get('condim', mix['condim'])} is not implemented (pyramidal cones with condim 1, 3, 4)")
            s.add_contact_pair(a["geom1"], a["geom2"], condim=int(a.get("condim", mix["condim"])),
                               friction=tuple(_floats(a.get("friction"), None, mix["friction"])[:3]),
                               margin=float(a.get("margin", mix["margin"])), gap=float(a.get("gap", mix["gap"])),
                               solref=tuple(_floats(a.get("solref"), 2, mix["solref"])),
                               solimp=tuple(_floats(a.get("solimp"), 5, mix["solimp"])))
            explicit.add(frozenset((a["geom1"], a["geom2"])))
    bparent = {b.name: s.bodies[b.parent].name if b.parent >= 0 else None for b in s.bodies}
    welded_to_world = {b.name for b in s.bodies if not _has_dofs_to_world(s, b.name)}
    for i, g1 in enumerate(geom_info):
        for g2 in geom_info[i + 1:]:
            if not ((g1["contype"] & g2["conaffinity"]) or (g2["contype"] & g1["conaffinity"])):
                continue
            b1, b2 = g1["body"], g2["body"]
            if b1 == b2 or frozenset((b1, b2)) in excl or frozenset((g1["name"], g2["name"])) in explicit:
                continue
            if b1 in welded_to_world and b2 in welded_to_world:
                continue                                                  # two static geoms never collide
            if (bparent.get(b1) == b2 or bparent.get(b2) == b1) and b1 != "world" and b2 != "world":
                if not (bparent.get(b1) == b2 and b2 in welded_to_world) and not (bparent.get(b2) == b1 and b1 in welded_to_world):
                    continue                                              # parent-child filter
            mix = _mix(g1, g2)
            if mix["condim"] not in (1, 3, 4):
                raise MjcfError(f"contact pair {g1['name']} / {g2['name']}: condim {mix['condim']} is not implemented (pyramidal cones with condim 1, 3, 4)")
            s.add_contact_pair(g1["name"], g2["name"], condim=mix["condim"], friction=tuple(mix["friction"]),
                               margin=mix["margin"], gap=mix["gap"], solref=tuple(mix["solref"]), solimp=tuple(mix["solimp"]))

    # ---- keyframes
    keys = []
    for kf in root.findall("keyframe"):
        for el in kf.findall("key"):
            keys.append((el.attrib.get("qpos"), el.attrib.get("qvel")))
    if keys:
        nq = sum({C["MM_JNT_FREE"]: 7, C["MM_JNT_BALL"]: 4}.get(j.type, 1) for j in s.joints)
        nv = sum({C["MM_JNT_FREE"]: 6, C["MM_JNT_BALL"]: 3}.get(j.type, 1) for j in s.joints)
        if ctx.lenient:      # keyframes are sized for the whole model: with includes skipped they cannot match what was read
            ok = [(q, v) for q, v in keys if (q is None or len(q.split()) == nq) and (v is None or len(v.split()) == nv)]
            if len(ok) != len(keys):
                ctx.skipped["keyframes sized for the complete model"] = len(keys) - len(ok)
            keys = ok
        s.keys = [(np.array(_floats(q, nq)) if q else None, np.array(_floats(v, nv)) if v else np.zeros(nv)) for q, v in keys]
    s.import_skipped = dict(ctx.skipped)
    return s


def _has_dofs_to_world(s: ModelSpec, body: str) -> bool:
    b = s._bname[body]
    while b > 0:
        if s.bodies[b].joints:
            return True
        b = s.bodies[b].parent
    return False


def _mix(g1, g2):
    """contact parameters of a generated geom pair (MuJoCo: higher priority wins, else max condim / friction / margin / gap,
    solmix-weighted solref / solimp)"""
    if g1["priority"] != g2["priority"]:
        g = g1 if g1["priority"] > g2["priority"] else g2
        return dict(condim=g["condim"], friction=list(g["friction"]), margin=max(g1["margin"], g2["margin"]),
                    gap=max(g1["gap"], g2["gap"]), solref=list(g["solref"]), solimp=list(g["solimp"]))
    w1 = g1["solmix"] / max(1e-15, g1["solmix"] + g2["solmix"]) if (g1["solmix"] + g2["solmix"]) > 0 else 0.5
    sr = [w1 * a + (1 - w1) * b for a, b in zip(g1["solref"], g2["solref"])]
    if g1["solref"][0] <= 0 or g2["solref"][0] <= 0:          # direct (stiffness, damping) format: MuJoCo takes the minimum
        sr = [min(a, b) for a, b in zip(g1["solref"], g2["solref"])]
    return dict(condim=max(g1["condim"], g2["condim"]), friction=[max(a, b) for a, b in zip(g1["friction"], g2["friction"])],
                margin=max(g1["margin"], g2["margin"]), gap=max(g1["gap"], g2["gap"]), solref=sr,
                solimp=[w1 * a + (1 - w1) * b for a, b in zip(g1["solimp"], g2["solimp"])])


# ----------------------------------------------------------------------------- exporter
def _f(v):
    return " ".join(repr(float(x)) for x in np.asarray(v, float).reshape(-1))


_GT = {0: "plane", 2: "sphere", 3: "capsule", 4: "ellipsoid", 5: "cylinder", 6: "box"}


def dry_run(source: str, include_map: Optional[Dict[str, str]] = None) -> dict:
    """What would `load` reject?  Walks an MJCF file (includes expanded where they resolve, recorded where they do not -- the
    reference's task XMLs point into the empty ``simhive/myo_sim`` submodule) and lists EVERY construct outside the subset the
    engine implements instead of stopping at the first, so that whoever gets hold of the real ``myo_sim`` tree sees the whole
    gap at once.  Returns {"file", "missing_includes": [...], "unsupported": {construct: [locations...]}, "ignored": {...},
    "counts": {tag: n}, "loadable": bool}.  `unsupported` mirrors the MjcfError sites of `load`; `ignored` lists elements that
    `load` skips on purpose because they do not touch the physics step (visuals, cameras, lights, sensors, keyframes beyond qpos /
    qvel / act / ctrl)."""
    if include_map is None and os.environ.get("MYOSUITE_MYO_SIM_ROOT"):
        include_map = {"simhive/myo_sim": os.environ["MYOSUITE_MYO_SIM_ROOT"]}
    if os.path.exists(source):
        root = ET.parse(source).getroot(); base = os.path.dirname(os.path.abspath(source)); fname = source
    else:
        root = ET.fromstring(source); base = os.getcwd(); fname = "<string>"
    missing: List[str] = []

    def note_missing(node, b):
        for ch in list(node):
            if ch.tag == "include":
                fn = ch.attrib.get("file", "")
                for old, new_ in (include_map or {}).items():
                    if old in fn:
                        fn = os.path.join(new_, fn[fn.index(old) + len(old):].lstrip("/"))
                        break
                path = fn if os.path.isabs(fn) else os.path.join(b, fn)
                if not os.path.exists(path):
                    missing.append(ch.attrib.get("file", ""))
            else:
                note_missing(ch, b)
    note_missing(root, base)
    _expand_includes(root, base, "skip", include_map)
    unsupported: Dict[str, List[str]] = {}
    ignored: Dict[str, int] = {}
    counts: Dict[str, int] = {}

    def bad(kind, where):
        unsupported.setdefault(kind, []).append(where)

    def where(el, parents):
        nm = el.attrib.get("name") or el.attrib.get("class") or ""
        return "/".join(p.tag + (f"[{p.attrib.get('name')}]" if p.attrib.get("name") else "") for p in parents[-2:]) + f"/{el.tag}" + (f"[{nm}]" if nm else "")

    ctx = _Ctx(); ctx.lenient = True
    for dn in root.findall("default"):
        ctx.read_defaults(dn)

    def childclass_of(parents):
        cc = None
        for p in parents:
            if p.tag in ("body", "worldbody", "frame") and "childclass" in p.attrib:
                cc = p.attrib["childclass"]
        return cc

    def walk(el, parents):
        counts[el.tag] = counts.get(el.tag, 0) + 1
        a, w = el.attrib, where(el, parents)
        t = el.tag
        top = parents[1].tag if len(parents) > 1 else (parents[0].tag if parents else "")
        if t == "compiler" and a.get("coordinate", "local") != "local":
            bad("compiler coordinate=global", w)
        if t == "option":
            if a.get("integrator", "Euler") not in ("Euler", "RK4", "implicitfast"):
                bad(f"integrator {a['integrator']}", w)
            if a.get("cone", "pyramidal") != "pyramidal":
                bad("elliptic friction cone", w)
            if a.get("solver", "Newton") != "Newton":
                bad(f"solver {a['solver']}", w)
        if t == "geom" and "default" not in [p.tag for p in parents]:
            ra = ctx.resolve("geom", dict(a), childclass_of(parents))       # the element's attributes through its default classes
            gtype = ra.get("type")
            if gtype in ("mesh", "hfield", "sdf") or ("mesh" in ra and gtype is None):
                # collides unless contype == conaffinity == 0 (MuJoCo's defaults are 1 / 1)
                if int(float(ra.get("contype", "1"))) == 0 and int(float(ra.get("conaffinity", "1"))) == 0:
                    ignored["visual mesh geom"] = ignored.get("visual mesh geom", 0) + 1
                else:
                    bad(f"{gtype or 'mesh'} geom that collides (contype {ra.get('contype', '1')} / conaffinity {ra.get('conaffinity', '1')} after default classes)", w)
            elif int(float(ra.get("condim", "3"))) not in (1, 3, 4) and not (int(float(ra.get("contype", "1"))) == 0 and int(float(ra.get("conaffinity", "1"))) == 0):
                bad(f"colliding geom with condim {ra.get('condim')} (pyramidal cones are implemented for condim 1, 3 and 4: no rolling friction)", w)
        if t == "joint" and "default" not in [p.tag for p in parents]:
            if a.get("type") == "ball" and (a.get("limited") == "true" or "range" in a):
                bad("limited ball joint", w)
        if t in ("weld", "connect", "tendon", "distance", "flex") and top == "equality":
            bad(f"equality <{t}>", w)
        if top == "actuator" and t in ("general", "motor", "position", "velocity", "muscle", "cylinder", "adhesion", "damper", "intvelocity", "plugin"):
            if not ("joint" in a or "tendon" in a):
                bad("actuator transmission other than joint / tendon (" + ", ".join(k for k in ("site", "body", "slidersite", "cranksite", "jointinparent") if k in a) + ")", w)
            if t in ("cylinder", "adhesion", "plugin"):
                bad(f"actuator <{t}>", w)
            if t == "general" and a.get("dyntype", "none") not in ("none", "integrator", "filter", "filterexact", "muscle"):
                bad(f"actuator dyntype {a['dyntype']}", w)
        if t in ("composite", "flexcomp", "flex", "skin", "plugin", "extension", "replicate", "attach", "frame") and t != "frame":
            bad(f"<{t}>", w)
        if t == "spatial" and top == "tendon":
            for ch in el:
                if ch.tag == "pulley":
                    pass   # pulleys are implemented (divisor)
        if t in ("camera", "light", "visual", "asset", "texture", "material", "sensor", "custom", "statistic", "headlight", "rgba", "global", "quality", "map", "scale"):
            ignored[t] = ignored.get(t, 0) + 1
            if t in ("asset", "sensor", "visual", "custom"):
                return
        for ch in el:
            walk(ch, parents + [el])
    walk(root, [])
    return {"file": fname, "missing_includes": missing, "unsupported": unsupported, "ignored": ignored, "counts": counts,
            "loadable": not unsupported and not missing}


def dump(spec: ModelSpec) -> str:
    """ModelSpec -> MJCF text (radians, explicit inertials, collision through explicit <pair>s only)."""
    root = ET.Element("mujoco", model=spec.name)
    ET.SubElement(root, "compiler", angle="radian", autolimits="true", inertiafromgeom="false")
    o = ET.SubElement(root, "option", timestep=repr(float(spec.timestep)), gravity=_f(spec.gravity),
                      integrator={0: "Euler", 1: "RK4", 3: "implicitfast"}[spec.integrator], iterations=str(spec.iterations),
                      tolerance=repr(float(spec.tolerance)), ls_iterations=str(spec.ls_iterations),
                      ls_tolerance=repr(float(spec.ls_tolerance)), cone="pyramidal", solver="Newton")
    if not spec.eulerdamp:
        ET.SubElement(o, "flag", eulerdamp="disable")
    if spec.nconmax or getattr(spec, "njmax", 0):
        ET.SubElement(root, "size", **({"nconmax": str(spec.nconmax)} if spec.nconmax else {}),
                      **({"njmax": str(spec.njmax)} if getattr(spec, "njmax", 0) else {}))
    wb = ET.SubElement(root, "worldbody")
    nodes = {0: wb}
    jt = {C["MM_JNT_FREE"]: "free", C["MM_JNT_BALL"]: "ball", C["MM_JNT_SLIDE"]: "slide", C["MM_JNT_HINGE"]: "hinge"}

    def emit_attached(bi, node):
        for g in spec.geoms:
            if g["body"] == bi:
                ET.SubElement(node, "geom", name=g["name"], type=_GT[g["type"]], size=_f(g["size"]), pos=_f(g["pos"]),
                              quat=_f(g["quat"]), contype="0", conaffinity="0")
        for n, b, p in spec.sites:
            if b == bi:
                ET.SubElement(node, "site", name=n, pos=_f(p))
    emit_attached(0, wb)
    for bi, b in enumerate(spec.bodies):
        if bi == 0:
            continue
        node = ET.SubElement(nodes[b.parent], "body", name=b.name, pos=_f(b.pos), quat=_f(b.quat))
        nodes[bi] = node
        if b.mass > 0:
            ET.SubElement(node, "inertial", pos=_f(b.ipos), quat=_f(b.iquat), mass=repr(float(b.mass)), diaginertia=_f(b.inertia))
        for ji in b.joints:
            j = spec.joints[ji]
            if j.type == C["MM_JNT_FREE"]:
                ET.SubElement(node, "freejoint", name=j.name)
                continue
            at = dict(name=j.name, type=jt[j.type], pos=_f(j.pos), axis=_f(j.axis), stiffness=repr(j.stiffness),
                      damping=repr(j.damping), armature=repr(j.armature), ref=repr(j.ref), springref=repr(j.springref),
                      margin=repr(j.margin), solreflimit=_f(j.solref), solimplimit=_f(j.solimp),
                      limited="true" if j.limited else "false")
            if j.limited:
                at["range"] = _f(j.range)
            if j.frictionloss > 0:
                at.update(frictionloss=repr(j.frictionloss), solreffriction=_f(j.solreffriction), solimpfriction=_f(j.solimpfriction))
            ET.SubElement(node, "joint", **at)
        emit_attached(bi, node)
    if spec.tendons:
        tn = ET.SubElement(root, "tendon")
        for t in spec.tendons:
            fixed = any(p[0] == "joint" for p in t.path)
            el = ET.SubElement(tn, "fixed" if fixed else "spatial", name=t.name, stiffness=repr(t.stiffness),
                               damping=repr(t.damping), limited="true" if t.limited else "false")
            if t.limited:
                el.set("range", _f(t.range)); el.set("margin", repr(t.margin))
                el.set("solreflimit", _f(t.solref)); el.set("solimplimit", _f(t.solimp))
            if t.springlength[0] >= 0:
                el.set("springlength", _f(t.springlength))
            for p in t.path:
                if p[0] == "site":
                    ET.SubElement(el, "site", site=p[1])
                elif p[0] in ("sphere", "cylinder"):
                    g = ET.SubElement(el, "geom", geom=p[1])
                    if p[2]:
                        g.set("sidesite", p[2])
                elif p[0] == "pulley":
                    ET.SubElement(el, "pulley", divisor=repr(float(p[1])))
                elif p[0] == "joint":
                    ET.SubElement(el, "joint", joint=p[1], coef=repr(float(p[2])))
    if spec.actuators:
        ac = ET.SubElement(root, "actuator")
        for a in spec.actuators:
            at = dict(name=a.name, gear=repr(a.gear), ctrllimited="true" if a.ctrllimited else "false",
                      forcelimited="true" if a.forcelimited else "false")
            at["tendon" if a.trntype == C["MM_TRN_TENDON"] else "joint"] = a.target
            if a.ctrllimited:
                at["ctrlrange"] = _f(a.ctrlrange)
            if a.forcelimited:
                at["forcerange"] = _f(a.forcerange)
            if a.lengthrange is not None:
                at["lengthrange"] = _f(a.lengthrange)
            if a.dyntype == C["MM_DYN_MUSCLE"]:
                at.update(dyntype="muscle", gaintype="muscle", biastype="muscle", dynprm=_f(a.dynprm), gainprm=_f(a.gainprm),
                          biasprm=_f(a.biasprm))
            else:
                dn = {C["MM_DYN_NONE"]: "none", C["MM_DYN_INTEGRATOR"]: "integrator", C["MM_DYN_FILTER"]: "filter"}[a.dyntype]
                at.update(dyntype=dn, gaintype="fixed", gainprm=_f(a.gainprm[:1]), dynprm=_f(a.dynprm))
                if a.biastype == C["MM_BIAS_AFFINE"]:
                    at.update(biastype="affine", biasprm=_f(a.biasprm[:3]))
                else:
                    at.update(biastype="none")
            ET.SubElement(ac, "general", **at)
    if spec.equalities:
        eq = ET.SubElement(root, "equality")
        for e in spec.equalities:
            at = dict(joint1=e["j1"], polycoef=_f(e["data"]), solref=_f(e["solref"]), solimp=_f(e["solimp"]))
            if e["j2"] is not None:
                at["joint2"] = e["j2"]
            ET.SubElement(eq, "joint", **at)
    if spec.pairs:
        ct = ET.SubElement(root, "contact")
        for p in spec.pairs:
            ET.SubElement(ct, "pair", geom1=p["g1"], geom2=p["g2"], condim=str(p["condim"]), friction=_f(p["friction"]),
                          margin=repr(p["margin"]), gap=repr(p["gap"]), solref=_f(p["solref"]), solimp=_f(p["solimp"]))
    keys = getattr(spec, "keys", None)
    if keys:
        kf = ET.SubElement(root, "keyframe")
        for q, v in keys:
            ET.SubElement(kf, "key", qpos=_f(q), qvel=_f(v))
    ET.indent(root)
    return ET.tostring(root, encoding="unicode")


def dump_tree(spec: ModelSpec, root_dir: str, family: str = "hand", assets: str = "myohand_assets.xml", body: str = "myohand_body.xml",
              drop_world_sites=()) -> Dict[str, str]:
    """Write ``spec`` as the include tree the reference's task XMLs expect from the ``myo_sim`` submodule
    (envs/myo/assets/hand/myohand_pose.xml:10-15): ``<root>/<family>/assets/<assets>`` (compiler / option / size / tendon /
    actuator / equality / contact), ``<root>/<family>/assets/<body>`` (the body tree, included inside <worldbody>) and an empty
    ``<root>/scene/myosuite_scene.xml``.  ``drop_world_sites``: world-attached sites the task XML authors itself (the *_target
    sites).  A stand-in for the (absent) real tree; returns the written paths."""
    root = ET.fromstring(dump(spec))
    wb = root.find("worldbody")
    body_inc = ET.Element("mujocoinclude")
    for ch in list(wb):
        if ch.tag == "site" and ch.attrib.get("name") in set(drop_world_sites):
            continue
        body_inc.append(ch)
    assets_inc = ET.Element("mujocoinclude")
    for ch in list(root):
        if ch.tag not in ("worldbody", "keyframe"):
            assets_inc.append(ch)
    out = {}
    for rel, el in ((os.path.join(family, "assets", assets), assets_inc), (os.path.join(family, "assets", body), body_inc),
                    (os.path.join("scene", "myosuite_scene.xml"), ET.Element("mujocoinclude"))):
        path = os.path.join(root_dir, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        ET.indent(el)
        with open(path, "w") as f:
            f.write(ET.tostring(el, encoding="unicode"))
        out[rel] = path
    return out
