"""A stand-in for the `mujoco` Python module with the handful of entry points tests/tools/validate_against_mujoco.py touches,
implemented ON THE fp64 ORACLE (test plumbing only: it lets the fixture writer / reader pair and the field mapping be exercised
on hosts without libmujoco, so that the first real run cannot fail on plumbing).  Numbers it produces are the oracle's own; a
fixture written through it says so (mujoco_version = "fake-oracle") and is never committed."""
import types

import numpy as np

__version__ = "fake-oracle"


class _Stat:
    def __init__(self, meaninertia):
        self.meaninertia = meaninertia


class MjModel:
    def __init__(self, cm):
        from oracle import oracle as O
        self._cm = cm
        self._om = O.OracleModel(cm)
        A = cm.arrays
        self.nq, self.nv, self.nu, self.na, self.ntendon, self.nbody = cm.nq, cm.nv, cm.nu, cm.na, cm.ntendon, cm.nbody
        self.dof_invweight0 = np.array(A["DOF_INVWEIGHT0"], np.float64)
        self.body_invweight0 = np.array(A["BODY_INVWEIGHT0"], np.float64).reshape(-1, 2)
        self.tendon_invweight0 = np.array(A["TENDON_INVWEIGHT0"], np.float64)
        self.actuator_acc0 = np.array(A["ACT_ACC0"], np.float64)
        self.actuator_lengthrange = np.array(A["ACT_LENGTHRANGE"], np.float64).reshape(-1, 2)
        self.stat = _Stat(float(A["OPT_F"][2]))

    @classmethod
    def from_xml_path(cls, path):
        from myosuite_amd.model import mjcf
        return cls(mjcf.load(path).compile())


class MjData:
    """attribute access goes to the oracle's arrays (same field names as mjData)"""

    def __init__(self, model):
        from oracle import oracle as O
        object.__setattr__(self, "_m", model)
        object.__setattr__(self, "_d", O.OracleData(model._om))

    def __getattr__(self, name):
        if name in ("nefc", "ncon"):
            return getattr(self._d, name)
        if name == "contact":           # mjData.contact[i].dist / .pos / .frame (normal first), in detection order
            d = self._d
            return [types.SimpleNamespace(dist=float(d.con_dist[i]), pos=np.array(d.con_pos[i]), frame=np.array(d.con_frame[i]))
                    for i in range(d.ncon)]
        return getattr(self._d, name)


def mj_forward(m, d):
    d._d.forward()


def mj_step(m, d):
    d._d.step()


def install():
    """register this module as `mujoco` (tests only)"""
    import sys
    mod = types.ModuleType("mujoco")
    for k in ("__version__", "MjModel", "MjData", "mj_forward", "mj_step"):
        setattr(mod, k, globals()[k])
    sys.modules["mujoco"] = mod
    return mod
