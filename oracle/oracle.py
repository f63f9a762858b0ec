"""ctypes binding of the fp64 CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (myosuite_amd/) never imports
this module; its step path fails loudly when the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Which build of the SAME sources this process binds (oracle/Makefile): "" = the checker (-O2, no contraction: every parity test),
# "fast" = -O3 -march=native (bench.py's cpu_baseline leg only), "asan" = AddressSanitizer + UBSan (tests/test_oracle_sanitizers.py,
# in a child process with libasan preloaded).  Chosen once per process: $MYOSIM_ORACLE_VARIANT or use_variant() before the first call.
_VARIANTS = {"": "liboracle.so", "fast": "liboracle_fast.so", "asan": "liboracle_asan.so"}
_variant = os.environ.get("MYOSIM_ORACLE_VARIANT", "")
_LIB_PATH = os.path.join(_HERE, _VARIANTS[_variant])
_lib = None


def variant() -> str:
    return _variant


def use_variant(name: str) -> None:
    """Select the build this process binds; only before the first oracle call (one process, one library)."""
    global _variant, _LIB_PATH
    if name not in _VARIANTS:
        raise ValueError(f"unknown oracle variant {name!r}")
    if _lib is not None and name != _variant:
        raise RuntimeError(f"oracle library {_VARIANTS[_variant]} is already loaded in this process")
    _variant, _LIB_PATH = name, os.path.join(_HERE, _VARIANTS[name])


def _host_cpu() -> str:
    """identity of the CPU a -march=native build is valid for"""
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = hashlib.sha1(line.split(":", 1)[1].strip().encode()).hexdigest()[:8]    # stable across processes (str hash is salted)
            if model and flags:
                break
    except OSError:
        pass
    return f"{model}|{flags}"


def build(force: bool = False, variant: str = None) -> str:
    """make the selected variant (default: the one this process binds) when a source is newer than it; the -march=native variant is
    also rebuilt when it was built on another CPU (built .so files travel to the GPU box with the snapshot)."""
    v = _variant if variant is None else variant
    path = os.path.join(_HERE, _VARIANTS[v])
    srcs = [os.path.join(_HERE, f) for f in ("mmo_engine.c", "mmo_collision.inc", "mmo_batch.c", "Makefile")]
    srcs.append(os.path.join(_HERE, "..", "include", "myosim_model.h"))
    stamp = path + ".host"
    if v == "fast" and os.path.exists(path) and not (os.path.exists(stamp) and open(stamp).read() == _host_cpu()):
        force = True
    if force or not os.path.exists(path) or any(
            os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
        # built under a temporary name and moved into place: a concurrent process never dlopens a half-written library
        tmp = f".{os.getpid()}.{_VARIANTS[v]}"
        try:
            subprocess.check_call(["make", "-C", _HERE, "-B", _VARIANTS[v], f"OUT={tmp}"], stdout=subprocess.DEVNULL)
            os.replace(os.path.join(_HERE, tmp), path)
        finally:
            if os.path.exists(os.path.join(_HERE, tmp)):
                os.remove(os.path.join(_HERE, tmp))
        if v == "fast":
            with open(stamp, "w") as f:
                f.write(_host_cpu())
    return path


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.mmo_model_load.restype = C.c_void_p
        L.mmo_model_load.argtypes = [C.c_void_p, C.c_int]
        L.mmo_model_free.argtypes = [C.c_void_p]
        L.mmo_data_create.restype = C.c_void_p
        L.mmo_data_create.argtypes = [C.c_void_p]
        L.mmo_data_free.argtypes = [C.c_void_p]
        for f in ("mmo_reset", "mmo_forward", "mmo_step", "mmo_fwd_position", "mmo_fwd_velocity"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        L.mmo_field.restype = C.POINTER(C.c_double)
        L.mmo_field.argtypes = [C.c_void_p, C.c_char_p]
        L.mmo_time.restype = C.c_double
        L.mmo_time.argtypes = [C.c_void_p]
        L.mmo_set_time.argtypes = [C.c_void_p, C.c_double]
        L.mmo_set_geom_size.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.mmo_set_geom_type.argtypes = [C.c_void_p, C.c_int]
        L.mmo_set_round_state_f32.argtypes = [C.c_void_p, C.c_int]
        L.mmo_set_body_mass.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.mmo_set_body_pos.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        L.mmo_test_seg_shape.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_double, C.c_void_p, C.c_void_p]
        L.mmo_test_seg_shape.restype = C.c_double
        for f in ("mmo_nefc", "mmo_ncon", "mmo_solver_niter", "mmo_warn"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.mmo_efc_type.restype = C.POINTER(C.c_int)
        L.mmo_efc_type.argtypes = [C.c_void_p]
        L.mmo_con_pair.restype = C.POINTER(C.c_int)
        L.mmo_con_pair.argtypes = [C.c_void_p]
        L.mmo_full_m.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.mmo_batch_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int]
        _lib = L
    return _lib


class OracleModel:
    def __init__(self, compiled):
        """compiled: myosuite_amd.model.spec.CompiledModel (or anything with .blob + dims)"""
        self.cm = compiled
        blob = np.ascontiguousarray(compiled.blob, dtype=np.uint32)
        self._blob = blob
        self.ptr = lib().mmo_model_load(blob.ctypes.data, int(blob.size))
        if not self.ptr:
            raise RuntimeError("oracle rejected the model blob")

    def __del__(self):
        try:
            if self.ptr:
                lib().mmo_model_free(self.ptr)
        except Exception:
            pass


_SHAPES = {
    "qpos": lambda m: (m.nq,), "qvel": lambda m: (m.nv,), "act": lambda m: (m.na,), "ctrl": lambda m: (m.nu,),
    "qacc_warmstart": lambda m: (m.nv,), "xpos": lambda m: (m.nbody, 3), "xquat": lambda m: (m.nbody, 4),
    "xmat": lambda m: (m.nbody, 9), "xipos": lambda m: (m.nbody, 3), "ximat": lambda m: (m.nbody, 9),
    "xanchor": lambda m: (m.njnt, 3), "xaxis": lambda m: (m.njnt, 3), "site_xpos": lambda m: (m.nsite, 3),
    "geom_xpos": lambda m: (m.ngeom, 3), "geom_xmat": lambda m: (m.ngeom, 9),
    "subtree_com": lambda m: (m.nbody, 3), "cinert": lambda m: (m.nbody, 10), "cdof": lambda m: (m.nv, 6),
    "crb": lambda m: (m.nbody, 10), "ten_length": lambda m: (m.ntendon,), "ten_J": lambda m: (m.ntendon, m.nv),
    "actuator_length": lambda m: (m.nu,), "actuator_moment": lambda m: (m.nu, m.nv), "qM": lambda m: (m.nM,),
    "qLD": lambda m: (m.nM,), "qLDiagInv": lambda m: (m.nv,), "ten_velocity": lambda m: (m.ntendon,),
    "actuator_velocity": lambda m: (m.nu,), "cvel": lambda m: (m.nbody, 6), "cdof_dot": lambda m: (m.nv, 6),
    "qfrc_passive": lambda m: (m.nv,), "qfrc_bias": lambda m: (m.nv,), "act_dot": lambda m: (m.na,),
    "actuator_force": lambda m: (m.nu,), "qfrc_actuator": lambda m: (m.nv,), "qfrc_smooth": lambda m: (m.nv,),
    "qacc_smooth": lambda m: (m.nv,), "efc_J": lambda m: (max(m.njmax, 1), m.nv),
    "efc_pos": lambda m: (max(m.njmax, 1),), "efc_margin": lambda m: (max(m.njmax, 1),),
    "efc_R": lambda m: (max(m.njmax, 1),), "efc_D": lambda m: (max(m.njmax, 1),),
    "efc_vel": lambda m: (max(m.njmax, 1),), "efc_aref": lambda m: (max(m.njmax, 1),),
    "efc_force": lambda m: (max(m.njmax, 1),), "qfrc_constraint": lambda m: (m.nv,), "qacc": lambda m: (m.nv,),
    "cfrc": lambda m: (m.nbody, 6), "cacc": lambda m: (m.nbody, 6),
    "con_dist": lambda m: (max(m.nconmax, 1),), "con_pos": lambda m: (max(m.nconmax, 1), 3),
    "con_frame": lambda m: (max(m.nconmax, 1), 9), "efc_diagApprox": lambda m: (max(m.njmax, 1),),
    "efc_floss": lambda m: (max(m.njmax, 1),),
}


class OracleData:
    """One env's mjData-like state; attributes are numpy views into the C arrays."""

    def __init__(self, model: OracleModel):
        self.model = model
        self.ptr = lib().mmo_data_create(model.ptr)

    def __del__(self):
        try:
            if self.ptr:
                lib().mmo_data_free(self.ptr)
        except Exception:
            pass

    def __getattr__(self, name):
        if name in _SHAPES:
            shape = _SHAPES[name](self.model.cm)
            n = int(np.prod(shape))
            p = lib().mmo_field(self.ptr, name.encode())
            if n == 0:
                return np.zeros(shape)
            return np.ctypeslib.as_array(p, shape=(n,)).reshape(shape)
        raise AttributeError(name)

    @property
    def time(self):
        return lib().mmo_time(self.ptr)

    @time.setter
    def time(self, t):
        lib().mmo_set_time(self.ptr, float(t))

    @property
    def con_pair(self):
        """entry of the PAIR_* sections every detected contact belongs to"""
        n = self.ncon
        return np.ctypeslib.as_array(lib().mmo_con_pair(self.ptr), shape=(max(n, 1),))[:n].copy()

    @property
    def efc_type(self):
        """constraint kind of every active row (MM_CON_*: 0 equality, 1 joint limit, 2 tendon limit, 3 contact, 4 dof friction)"""
        n = self.nefc
        return np.ctypeslib.as_array(lib().mmo_efc_type(self.ptr), shape=(max(n, 1),))[:n].copy()

    @property
    def nefc(self):
        return lib().mmo_nefc(self.ptr)

    @property
    def ncon(self):
        return lib().mmo_ncon(self.ptr)

    @property
    def solver_niter(self):
        return lib().mmo_solver_niter(self.ptr)

    @property
    def warn(self):
        return lib().mmo_warn(self.ptr)

    def reset(self):
        lib().mmo_reset(self.model.ptr, self.ptr)

    def set_geom_size(self, geom: int, size, gtype: int = -1):
        """per-env model delta: collision size (and optionally type) of one geom (the reference writes
        mj_model.geom_size / geom_type at reset)"""
        lib().mmo_set_geom_size(self.ptr, int(geom), float(size[0]), float(size[1]), float(size[2]))
        lib().mmo_set_geom_type(self.ptr, int(gtype))

    def set_body_mass(self, body: int, mass: float):
        """per-env model delta (pose_v0.py:183): mass of one body; its inertia tensor is untouched"""
        lib().mmo_set_body_mass(self.ptr, int(body), float(mass))

    def set_body_pos(self, body: int, pos):
        """per-env model delta (key_turn_v0.py:164): frame position of one body in its parent"""
        lib().mmo_set_body_pos(self.ptr, int(body), float(pos[0]), float(pos[1]), float(pos[2]))

    def round_state_f32(self, on: bool = True):
        """the fp32-STATE twin: fp64 arithmetic, but qpos / qvel / act / qacc_warmstart are rounded to float after every
        mmo_step -- the accuracy floor of any engine that keeps its state in fp32"""
        lib().mmo_set_round_state_f32(self.ptr, int(bool(on)))

    def forward(self):
        lib().mmo_forward(self.model.ptr, self.ptr)

    def step(self, n: int = 1):
        for _ in range(n):
            lib().mmo_step(self.model.ptr, self.ptr)

    def full_M(self):
        nv = self.model.cm.nv
        out = np.zeros((nv, nv))
        lib().mmo_full_m(self.model.ptr, self.ptr, out.ctypes.data)
        return out


def batch_rollout(model: OracleModel, datas, actions: np.ndarray, nsub: int, nthreads: int = 1,
                  normalize: bool = True, do_forward: bool = True):
    """actions [nsteps, nenv, nu] float64; advances every env in `datas` in place."""
    actions = np.ascontiguousarray(actions, dtype=np.float64)
    nsteps, nenv, nu = actions.shape
    assert nenv == len(datas)
    arr = (C.c_void_p * nenv)(*[d.ptr for d in datas])
    lib().mmo_batch_rollout(model.ptr, arr, nenv, nthreads, nsub, nsteps, actions.ctypes.data,
                            int(normalize), int(do_forward))


def seg_shape(gtype: int, size, a, u, h: float):
    """test hook: (signed distance, t, outward normal) of the segment a + t u, |t| <= h, against a convex primitive"""
    size = np.ascontiguousarray(size, np.float64); a = np.ascontiguousarray(a, np.float64); u = np.ascontiguousarray(u, np.float64)
    t = C.c_double(); g = np.zeros(3)
    sd = lib().mmo_test_seg_shape(int(gtype), size.ctypes.data, a.ctypes.data, u.ctypes.data, float(h), C.byref(t), g.ctypes.data)
    return sd, t.value, g
