"""Scan a synthetic model for tendon-length discontinuities inside the joint-limit box (wrap branch flips)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myosuite_amd.model import synth, kin_np as K

def scan(name, n=20000, eps=2e-4, seed=0, margin=0.1):
    cm = synth.get_model(name)
    km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
    rng = np.random.default_rng(seed)
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    lo2, hi2 = lo - margin, hi + margin        # soft limits are exceeded by up to ~0.1 rad
    q = lo2 + (hi2 - lo2) * rng.random((n, cm.nq))
    dq = rng.standard_normal((n, cm.nq)); dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    L0 = km.tendon_length(q); L1 = km.tendon_length(q + eps * dq)
    jump = np.abs(L1 - L0)
    names = list(cm.names["tendon"])
    bad = {}
    for t in range(cm.ntendon):
        k = jump[:, t] > 10 * eps * 0.05     # moment arms are < 5 cm
        if k.any():
            bad[names[t]] = (int(k.sum()), float(jump[:, t].max()))
    return bad

if __name__ == "__main__":
    for name in sys.argv[1:] or ["elbow", "hand"]:
        synth._CACHE.clear()
        print(name, scan(name))


def scan_kinks(name, n=100000, eps=1e-3, seed=1, margin=0.12):
    """first-derivative jumps: compare directional derivatives on both sides of q"""
    cm = synth.get_model(name)
    km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
    rng = np.random.default_rng(seed)
    lo, hi = cm.jnt_range[:, 0].astype(float) - margin, cm.jnt_range[:, 1].astype(float) + margin
    q = lo + (hi - lo) * rng.random((n, cm.nq))
    dq = rng.standard_normal((n, cm.nq)); dq /= np.linalg.norm(dq, axis=1, keepdims=True)
    Lm, L0, Lp = km.tendon_length(q - eps * dq), km.tendon_length(q), km.tendon_length(q + eps * dq)
    kink = np.abs((Lp - L0) - (L0 - Lm)) / eps          # jump of the directional derivative (m per rad)
    names = list(cm.names["tendon"])
    out = {}
    for t in range(cm.ntendon):
        k = kink[:, t] > 2e-3      # smooth curvature gives ~ eps * d2L/dq2 ~ 1e-3 * 0.05
        if k.any():
            i = int(np.argmax(kink[:, t]))
            out[names[t]] = (int(k.sum()), float(kink[:, t].max()), np.round(q[i], 2).tolist())
    return out
