"""Independent evidence for the two oracle stages that tests/test_oracle_invariants.py pins least directly:

 * the constraint stage (row assembly -> Newton): the SAME convex problem solved by a structurally different method -- the
   box-constrained DUAL  min_f 1/2 f'(J M^-1 J' + R) f + f'(J a0 - aref),  f >= 0 on limit / contact rows, |f| <= frictionloss
   on friction rows, f free on equalities (the formulation MuJoCo's PGS solver works on), by scipy's bounded-variable
   least-squares active-set solver -- must give the oracle's qacc and row forces;
 * the bias force (RNE): c(q, v) from the Lagrangian, d/dt(dT/dv) - dT/dq + dV/dq with T = 1/2 v'M(q)v, by finite differences
   of the oracle's mass matrix and of the potential energy from its forward kinematics -- no recursive Newton-Euler involved.
"""
import numpy as np
import pytest

from myosuite_amd.model import synth
from oracle import oracle as O

EQ, LIMJ, LIMT, CONTACT, FRIC = 0, 1, 2, 3, 4


def _dual_solve(d, cm):
    """qacc and row forces from the box-constrained dual QP (scipy bvls); inputs: the oracle's M, J, D, aref, qacc_smooth"""
    from scipy.optimize import lsq_linear
    n = d.nefc
    M = d.full_M()
    J = d.efc_J[:n].copy(); D = d.efc_D[:n].copy(); aref = d.efc_aref[:n].copy()
    ty = d.efc_type; floss = d.efc_floss[:n].copy()
    a0 = d.qacc_smooth.copy()
    MinvJt = np.linalg.solve(M, J.T)
    Q = J @ MinvJt + np.diag(1.0 / D)
    c = J @ a0 - aref
    lo = np.where(ty == EQ, -np.inf, np.where(ty == FRIC, -floss, 0.0))
    hi = np.where(ty == FRIC, floss, np.inf)
    # 1/2 f'Qf + c'f = 1/2 |U f + U^-T c|^2 + const with Q = U'U
    U = np.linalg.cholesky(Q).T
    b = -np.linalg.solve(U.T, c)
    res = lsq_linear(U, b, bounds=(lo, hi), method="bvls", tol=1e-14, max_iter=2000)
    f = res.x
    return a0 + MinvJt @ f, f


def _check(d, cm, tag):
    qacc, f = _dual_solve(d, cm)
    scale = max(1.0, np.abs(d.qacc).max())
    assert np.abs(qacc - d.qacc).max() < 2e-6 * scale, (tag, np.abs(qacc - d.qacc).max(), scale)
    fs = max(1.0, np.abs(f).max())
    assert np.abs(f - d.efc_force[:d.nefc]).max() < 2e-6 * fs, (tag, np.abs(f - d.efc_force[:d.nefc]).max())
    return f


def test_constraint_stage_equals_the_dual_qp_on_limits_equalities_contacts_friction(oracle_lib):
    rng = np.random.default_rng(3)
    kinds = set()
    active_rows = 0
    # hand: joint limits (states pushed beyond both range ends)
    cm = synth.get_model("hand"); om = O.OracleModel(cm)
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    for trial in range(6):
        d = O.OracleData(om)
        d.qpos[:] = (lo - 0.15 * (hi - lo)) + 1.3 * (hi - lo) * rng.random(cm.nq)
        d.qvel[:] = rng.standard_normal(cm.nv) * 3; d.act[:] = rng.random(cm.na); d.ctrl[:] = rng.random(cm.nu)
        d.forward()
        assert d.nefc >= 1
        f = _check(d, cm, ("hand", trial)); kinds |= set(d.efc_type.tolist()); active_rows += int((f > 0).sum())
    # leg: knee equalities + foot contacts (pyramidal), keyframe poses pressed into the floor
    cm = synth.get_model("leg"); om = O.OracleModel(cm)
    for trial in range(6):
        d = O.OracleData(om)
        q = cm.key_qpos[(0, 2, 3)[trial % 3]].astype(float).copy()
        q[7:] += rng.uniform(-0.02, 0.02, cm.nq - 7); q[2] -= rng.uniform(0.01, 0.04)
        d.qpos[:] = q; d.qvel[:] = rng.standard_normal(cm.nv) * 0.5; d.act[:] = rng.random(cm.na) * 0.5; d.ctrl[:] = rng.random(cm.nu)
        d.forward()
        assert d.ncon >= 1 and d.nefc >= 14
        f = _check(d, cm, ("leg", trial)); kinds |= set(d.efc_type.tolist()); active_rows += int((np.abs(f) > 0).sum())
    # friction toy (friction-loss rows saturated and not), tendon-limit toy, contact toy, reorient (capsule-vs-convex contacts)
    six_row_contacts = 0
    for name in ("friction_toy", "tendon_limit_toy", "contact_toy", "hand_reorient", "hand_pen"):      # hand_pen: condim 4, six rows per contact
        cm = synth.get_model(name); om = O.OracleModel(cm)
        hit = 0
        for trial in range(8):
            d = O.OracleData(om)
            q = cm.qpos0.astype(float).copy()
            if name in ("hand_reorient", "hand_pen"):
                q[:-6] = 0; q[0] = -1.5; q[-4] -= rng.uniform(0.012, 0.02)      # palm up, object pressed into the palm
            elif name == "contact_toy":
                q[2] -= rng.uniform(0.0, 0.03); q[9] -= rng.uniform(0.0, 0.25)
            else:
                q += rng.uniform(-0.6, 0.9, cm.nq)
            d.qpos[:] = q; d.qvel[:] = rng.standard_normal(cm.nv) * (0.3 if name != "friction_toy" else 2.0)
            if name == "hand_pen":
                d.qvel[-3:] = rng.standard_normal(3) * 20.0                     # the pen spins: the torsional rows have something to resist
            if cm.na:
                d.act[:] = rng.random(cm.na)
            d.ctrl[:] = rng.uniform(-1, 1, cm.nu) if name == "friction_toy" else rng.random(cm.nu)
            d.forward()
            if d.nefc == 0:
                continue
            hit += 1
            _check(d, cm, (name, trial)); kinds |= set(d.efc_type.tolist())
            if name == "hand_pen":
                six_row_contacts += int(np.sum(d.efc_type == CONTACT)) // 6; assert int(np.sum(d.efc_type == CONTACT)) == 6 * d.ncon
        assert hit >= 3, name
    assert six_row_contacts >= 3
    assert kinds == {EQ, LIMJ, LIMT, CONTACT, FRIC}, kinds
    assert active_rows > 20


def _lagrangian_bias(cm, d, q, v, dofs, h=1e-6):
    """c_k = sum_ij dM_kj/dq_i v_i v_j - 1/2 v' dM/dq_k v + dV/dq_k for hinge / slide coordinates k (q_k' = v_k), central
    differences of the oracle's full mass matrix and of V(q) = -sum_b m_b g.xipos_b"""
    A = cm.arrays
    qadr = A["JNT_QPOSADR"]; dadr = A["JNT_DOFADR"]; jtype = A["JNT_TYPE"]
    mass = A["BODY_MASS"].astype(float); grav = np.array([A["OPT_F"][1], A["OPT_F"][2], A["OPT_F"][3]], float)
    hinge = [(int(qadr[j]), int(dadr[j])) for j in range(cm.njnt) if jtype[j] in (2, 3)]

    def MV(qq):
        d.qpos[:] = qq; d.qvel[:] = 0; d.forward()
        return d.full_M().copy(), -float(np.sum(mass[:, None] * d.xipos * grav[None, :]))
    dM, dV = {}, {}
    for qa, da in hinge:
        qp = q.copy(); qp[qa] += h; qm = q.copy(); qm[qa] -= h
        Mp, Vp = MV(qp); Mm, Vm = MV(qm)
        dM[da] = (Mp - Mm) / (2 * h); dV[da] = (Vp - Vm) / (2 * h)
    Mdot = sum(dM[da] * v[da] for _, da in hinge)          # velocities of non-hinge dofs are zero in the test states
    c = np.zeros(cm.nv)
    for k in dofs:
        c[k] = Mdot[k] @ v - 0.5 * v @ dM[k] @ v + dV[k]
    return c


@pytest.mark.parametrize("name", ["hand", "leg", "elbow"])
def test_rne_bias_equals_the_finite_difference_lagrangian(oracle_lib, name):
    cm = synth.get_model(name); om = O.OracleModel(cm)
    A = cm.arrays
    jtype = A["JNT_TYPE"]; dadr = A["JNT_DOFADR"]
    hinge_dofs = [int(dadr[j]) for j in range(cm.njnt) if jtype[j] in (2, 3)]
    rng = np.random.default_rng(5)
    worst = 0.0
    for trial in range(2):
        d = O.OracleData(om)
        if name == "leg":
            q = cm.key_qpos[2].astype(float).copy(); q[7:] += rng.uniform(-0.2, 0.2, cm.nq - 7)
            qq = q[3:7] + rng.standard_normal(4) * 0.1; q[3:7] = qq / np.linalg.norm(qq)    # tilted root: gravity couples into every joint
        else:
            lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
            q = lo + (hi - lo) * rng.random(cm.nq)
        v = np.zeros(cm.nv); v[hinge_dofs] = rng.standard_normal(len(hinge_dofs)) * 2.0      # free-root velocity stays zero
        ref = _lagrangian_bias(cm, d, q, v, hinge_dofs)
        d.qpos[:] = q; d.qvel[:] = v; d.forward()
        got = d.qfrc_bias.copy()
        scale = max(1e-3, np.abs(ref[hinge_dofs]).max())
        err = np.abs(got[hinge_dofs] - ref[hinge_dofs]).max() / scale
        worst = max(worst, err)
    assert worst < 5e-6, worst
