"""Host-side (numpy, batched over sample configurations) kinematics used ONLY by
the model compiler in spec.py -- the counterpart of the MuJoCo compiler's
`set0` / length-range stage, which the reference reaches through
``MjSpec.compile()`` (myosuite/envs/env_base.py:72).  It is setup-time code: it
derives constants that are baked into the model blob (invweight0, meaninertia,
muscle lengthrange/acc0).  The per-step pipeline never calls into this module.

Semantics follow MuJoCo's documented kinematics / spatial-tendon wrapping
(SURVEY.md Appendix A1-A2).
"""
from __future__ import annotations

import numpy as np

from . import blob as _blob

C = _blob.C
MINVAL = 1e-15


def quat_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw], axis=-1)


def quat2mat(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    m = np.empty(q.shape[:-1] + (3, 3))
    m[..., 0, 0] = w * w + x * x - y * y - z * z
    m[..., 1, 1] = w * w - x * x + y * y - z * z
    m[..., 2, 2] = w * w - x * x - y * y + z * z
    m[..., 0, 1] = 2 * (x * y - w * z); m[..., 1, 0] = 2 * (x * y + w * z)
    m[..., 0, 2] = 2 * (x * z + w * y); m[..., 2, 0] = 2 * (x * z - w * y)
    m[..., 1, 2] = 2 * (y * z - w * x); m[..., 2, 1] = 2 * (y * z + w * x)
    return m


def axisangle_quat(axis, angle):
    h = 0.5 * angle
    s = np.sin(h)
    return np.concatenate([np.cos(h)[..., None], axis * s[..., None]], axis=-1)


def normalize(v):
    n = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.maximum(n, MINVAL)


class KinModel:
    def __init__(self, A, nq, nv, nbody):
        self.A = A; self.nq = nq; self.nv = nv; self.nbody = nbody
        f = lambda k, *shape: A[k].astype(np.float64).reshape(*shape) if A[k].size else np.zeros(shape)
        self.body_pos = f("BODY_POS", nbody, 3); self.body_quat = f("BODY_QUAT", nbody, 4)
        self.body_ipos = f("BODY_IPOS", nbody, 3); self.body_iquat = f("BODY_IQUAT", nbody, 4)
        self.body_mass = f("BODY_MASS", nbody); self.body_inertia = f("BODY_INERTIA", nbody, 3)
        self.parent = A["BODY_PARENT"]
        self.njnt = A["JNT_TYPE"].size
        self.jnt_pos = f("JNT_POS", self.njnt, 3); self.jnt_axis = f("JNT_AXIS", self.njnt, 3)
        self.qpos0 = A["QPOS0"].astype(np.float64)

    # ------------------------------------------------------------------ FK
    def fk(self, q):
        A = self.A
        S = q.shape[0]; nb = self.nbody
        xpos = np.zeros((S, nb, 3)); xquat = np.zeros((S, nb, 4)); xquat[:, 0, 0] = 1.0
        xanchor = np.zeros((S, self.njnt, 3)); xaxis = np.zeros((S, self.njnt, 3))
        for b in range(1, nb):
            p = self.parent[b]
            Rp = quat2mat(xquat[:, p])
            pos = xpos[:, p] + np.einsum("sij,j->si", Rp, self.body_pos[b])
            quat = quat_mul(xquat[:, p], np.broadcast_to(self.body_quat[b], (S, 4)))
            ja = A["BODY_JNTADR"][b]
            for j in range(ja, ja + A["BODY_JNTNUM"][b]) if ja >= 0 else []:
                t = A["JNT_TYPE"][j]; qa = A["JNT_QPOSADR"][j]
                if t == C["MM_JNT_FREE"]:
                    pos = q[:, qa:qa + 3].copy(); quat = normalize(q[:, qa + 3:qa + 7])
                    xanchor[:, j] = pos; xaxis[:, j] = quat2mat(quat)[:, :, 2]
                    continue
                R = quat2mat(quat)
                anchor = pos + np.einsum("sij,j->si", R, self.jnt_pos[j])
                axis = np.einsum("sij,j->si", R, self.jnt_axis[j])
                xanchor[:, j] = anchor; xaxis[:, j] = axis
                if t == C["MM_JNT_SLIDE"]:
                    pos = pos + axis * (q[:, qa] - self.qpos0[qa])[:, None]
                elif t == C["MM_JNT_HINGE"]:
                    ql = axisangle_quat(np.broadcast_to(self.jnt_axis[j], (S, 3)), q[:, qa] - self.qpos0[qa])
                    quat = quat_mul(quat, ql)
                    pos = anchor - np.einsum("sij,j->si", quat2mat(quat), self.jnt_pos[j])
                elif t == C["MM_JNT_BALL"]:
                    quat = quat_mul(quat, normalize(q[:, qa:qa + 4]))
                    pos = anchor - np.einsum("sij,j->si", quat2mat(quat), self.jnt_pos[j])
            xpos[:, b] = pos; xquat[:, b] = normalize(quat)
        return xpos, xquat, xanchor, xaxis

    def _dof_axes(self, xpos, xquat, xanchor, xaxis):
        """per dof: (kind, axis[S,3], point[S,3]); kind 0=translation, 1=rotation about point"""
        A = self.A
        out = []
        for j in range(self.njnt):
            t = A["JNT_TYPE"][j]; b = A["JNT_BODYID"][j]
            if t == C["MM_JNT_FREE"]:
                for k in range(3):
                    e = np.zeros_like(xpos[:, b]); e[:, k] = 1.0
                    out.append((0, e, None))
                R = quat2mat(xquat[:, b])
                for k in range(3):
                    out.append((1, R[:, :, k], xpos[:, b]))
            elif t == C["MM_JNT_BALL"]:
                R = quat2mat(xquat[:, b])
                for k in range(3):
                    out.append((1, R[:, :, k], xanchor[:, j]))
            elif t == C["MM_JNT_SLIDE"]:
                out.append((0, xaxis[:, j], None))
            else:
                out.append((1, xaxis[:, j], xanchor[:, j]))
        return out

    def _ancestor_dofs(self, b):
        A = self.A
        out = []
        while b > 0:
            n = A["BODY_DOFNUM"][b]
            if n > 0:
                out.extend(range(A["BODY_DOFADR"][b], A["BODY_DOFADR"][b] + n))
            b = self.parent[b]
        return out

    def point_jacobian(self, fkres, body, point):
        """Jp[S,3,nv], Jr[S,3,nv] of a world point attached to `body`."""
        xpos, xquat, xanchor, xaxis = fkres
        S = xpos.shape[0]
        Jp = np.zeros((S, 3, self.nv)); Jr = np.zeros((S, 3, self.nv))
        axes = self._dof_axes(xpos, xquat, xanchor, xaxis)
        for d in self._ancestor_dofs(body):
            kind, ax, pt = axes[d]
            if kind == 0:
                Jp[:, :, d] = ax
            else:
                Jp[:, :, d] = np.cross(ax, point - pt)
                Jr[:, :, d] = ax
        return Jp, Jr

    def body_com_jacobians(self, q):
        fkres = self.fk(q)
        xpos, xquat = fkres[0], fkres[1]
        S = q.shape[0]
        Jp = np.zeros((S, self.nbody, 3, self.nv)); Jr = np.zeros((S, self.nbody, 3, self.nv))
        for b in range(1, self.nbody):
            xipos = xpos[:, b] + np.einsum("sij,j->si", quat2mat(xquat[:, b]), self.body_ipos[b])
            Jp[:, b], Jr[:, b] = self.point_jacobian(fkres, b, xipos)
        return Jp, Jr

    def mass_matrix(self, q):
        S = q.shape[0]
        M = np.zeros((S, self.nv, self.nv))
        if self.nv == 0:
            return M
        fkres = self.fk(q)
        xpos, xquat = fkres[0], fkres[1]
        for b in range(1, self.nbody):
            m = self.body_mass[b]
            if m <= 0 and not np.any(self.body_inertia[b] > 0):
                continue
            R = quat2mat(xquat[:, b])
            xipos = xpos[:, b] + np.einsum("sij,j->si", R, self.body_ipos[b])
            Ri = np.einsum("sij,jk->sik", R, quat2mat(self.body_iquat[b]))
            Iw = np.einsum("sij,j,skj->sik", Ri, self.body_inertia[b], Ri)
            Jp, Jr = self.point_jacobian(fkres, b, xipos)
            M += m * np.einsum("ski,skj->sij", Jp, Jp)
            M += np.einsum("ski,skl,slj->sij", Jr, Iw, Jr)
        return M

    # ------------------------------------------------------------ tendons
    def site_xpos(self, fkres):
        xpos, xquat = fkres[0], fkres[1]
        sb = self.A["SITE_BODYID"]
        if sb.size == 0:
            return np.zeros((xpos.shape[0], 0, 3))
        sp = self.A["SITE_POS"].astype(np.float64).reshape(-1, 3)
        R = quat2mat(xquat[:, sb])
        return xpos[:, sb] + np.einsum("snij,nj->sni", R, sp)

    def geom_pose(self, fkres):
        xpos, xquat = fkres[0], fkres[1]
        gb = self.A["GEOM_BODYID"]
        if gb.size == 0:
            return np.zeros((xpos.shape[0], 0, 3)), np.zeros((xpos.shape[0], 0, 3, 3))
        gp = self.A["GEOM_POS"].astype(np.float64).reshape(-1, 3)
        gq = self.A["GEOM_QUAT"].astype(np.float64).reshape(-1, 4)
        R = quat2mat(xquat[:, gb])
        gx = xpos[:, gb] + np.einsum("snij,nj->sni", R, gp)
        gm = quat2mat(quat_mul(xquat[:, gb], np.broadcast_to(gq, xquat[:, gb].shape)))
        return gx, gm

    def tendon_length(self, q):
        A = self.A
        S = q.shape[0]
        nt = A["TENDON_ADR"].size
        L = np.zeros((S, nt))
        if nt == 0:
            return L
        fkres = self.fk(q)
        sx = self.site_xpos(fkres)
        gx, gm = self.geom_pose(fkres)
        wt, wo, wp = A["WRAP_TYPE"], A["WRAP_OBJID"], A["WRAP_PRM"]
        for t in range(nt):
            a0 = A["TENDON_ADR"][t]; n = A["TENDON_NUM"][t]
            div = 1.0
            j = 0
            while j < n - 1:
                t0 = wt[a0 + j]; t1 = wt[a0 + j + 1]
                if t0 == C["MM_WRAP_JOINT"]:
                    qa = A["JNT_QPOSADR"][wo[a0 + j]]
                    L[:, t] += wp[a0 + j] * q[:, qa]
                    j += 1
                    continue
                if t0 == C["MM_WRAP_PULLEY"] or t1 == C["MM_WRAP_PULLEY"]:
                    if t0 == C["MM_WRAP_PULLEY"]:
                        div = float(wp[a0 + j])
                    j += 1
                    continue
                p0 = sx[:, wo[a0 + j]]
                if t1 == C["MM_WRAP_SITE"]:
                    p1 = sx[:, wo[a0 + j + 1]]
                    L[:, t] += np.linalg.norm(p1 - p0, axis=-1) / div
                    j += 1
                else:
                    g = wo[a0 + j + 1]
                    p1 = sx[:, wo[a0 + j + 2]]
                    sid = int(round(float(wp[a0 + j + 1])))
                    side = sx[:, sid] if sid >= 0 else None
                    wlen, w0, w1 = wrap(p0, p1, gx[:, g], gm[:, g], float(A["GEOM_SIZE"][g][0]),
                                        t1 == C["MM_WRAP_CYLINDER"], side)
                    straight = np.linalg.norm(p1 - p0, axis=-1)
                    wrapped = np.linalg.norm(w0 - p0, axis=-1) + wlen + np.linalg.norm(p1 - w1, axis=-1)
                    L[:, t] += np.where(wlen < 0, straight, wrapped) / div
                    j += 2
            if n >= 1 and wt[a0 + n - 1] == C["MM_WRAP_JOINT"]:
                qa = A["JNT_QPOSADR"][wo[a0 + n - 1]]
                L[:, t] += wp[a0 + n - 1] * q[:, qa]
        return L

    def tendon_jacobian_fd(self, q, eps=1e-6):
        """Central-difference Jacobian of tendon lengths w.r.t. hinge/slide dofs
        (compile time only; free/ball dofs are left at zero: tendons of the
        shipped models never depend on them)."""
        A = self.A
        S = q.shape[0]; nt = A["TENDON_ADR"].size
        J = np.zeros((S, nt, self.nv))
        for j in range(self.njnt):
            if A["JNT_TYPE"][j] in (C["MM_JNT_SLIDE"], C["MM_JNT_HINGE"]):
                qa = A["JNT_QPOSADR"][j]; d = A["JNT_DOFADR"][j]
                qp = q.copy(); qp[:, qa] += eps
                qm = q.copy(); qm[:, qa] -= eps
                J[:, :, d] = (self.tendon_length(qp) - self.tendon_length(qm)) / (2 * eps)
        return J


# ---------------------------------------------------------------------- wrapping
def _is_intersect(p1, p2, p3, p4):
    det = (p4[:, 1] - p3[:, 1]) * (p2[:, 0] - p1[:, 0]) - (p4[:, 0] - p3[:, 0]) * (p2[:, 1] - p1[:, 1])
    n12 = (p2[:, 0] - p1[:, 0]) ** 2 + (p2[:, 1] - p1[:, 1]) ** 2
    n34 = (p4[:, 0] - p3[:, 0]) ** 2 + (p4[:, 1] - p3[:, 1]) ** 2
    ok = (np.abs(det) >= MINVAL) & (det * det >= 4e-6 * n12 * n34)   # (nearly) parallel segments never cross
    sd = np.where(ok, det, 1.0)
    a = ((p4[:, 0] - p3[:, 0]) * (p1[:, 1] - p3[:, 1]) - (p4[:, 1] - p3[:, 1]) * (p1[:, 0] - p3[:, 0])) / sd
    b = ((p2[:, 0] - p1[:, 0]) * (p1[:, 1] - p3[:, 1]) - (p2[:, 1] - p1[:, 1]) * (p1[:, 0] - p3[:, 0])) / sd
    return ok & (a >= 0) & (a <= 1) & (b >= 0) & (b <= 1)


def wrap_circle(d0, d1, sd, radius):
    """2-D wrap of segment d0->d1 around the origin-centred circle.
    returns wlen[S] (-1: no wrap), t0[S,2], t1[S,2] tangent points."""
    S = d0.shape[0]
    sqlen0 = np.sum(d0 * d0, -1); sqlen1 = np.sum(d1 * d1, -1); sqrad = radius * radius
    dif = d1 - d0
    dd = np.sum(dif * dif, -1)
    a = np.clip(-np.sum(dif * d0, -1) / np.maximum(MINVAL, dd), 0.0, 1.0)
    tmp = d0 + a[:, None] * dif
    nowrap = np.sum(tmp * tmp, -1) > sqrad
    if sd is not None:
        nowrap &= (np.sum(sd * tmp, -1) >= 0)
    nowrap |= (sqlen0 < sqrad) | (sqlen1 < sqrad)
    sqrt0 = np.sqrt(np.maximum(sqlen0 - sqrad, 0.0)); sqrt1 = np.sqrt(np.maximum(sqlen1 - sqrad, 0.0))
    s0 = np.maximum(sqlen0, MINVAL); s1 = np.maximum(sqlen1, MINVAL)
    sols = []; goods = []
    for sgn in (1.0, -1.0):
        t0 = np.stack([(d0[:, 0] * sqrad + sgn * radius * d0[:, 1] * sqrt0) / s0,
                       (d0[:, 1] * sqrad - sgn * radius * d0[:, 0] * sqrt0) / s0], -1)
        t1 = np.stack([(d1[:, 0] * sqrad - sgn * radius * d1[:, 1] * sqrt1) / s1,
                       (d1[:, 1] * sqrad + sgn * radius * d1[:, 0] * sqrt1) / s1], -1)
        if sd is not None:
            good = np.sum(normalize(t0 + t1) * sd, -1)
        else:
            good = -np.sum((t0 - t1) ** 2, -1)
        good = np.where(_is_intersect(d0, t0, d1, t1), -10000.0, good)
        sols.append((t0, t1)); goods.append(good)
    pick0 = goods[0] > goods[1]
    t0 = np.where(pick0[:, None], sols[0][0], sols[1][0])
    t1 = np.where(pick0[:, None], sols[0][1], sols[1][1])
    nowrap |= _is_intersect(d0, t0, d1, t1)
    cosang = np.clip(np.sum(t0 * t1, -1) / sqrad, -1.0, 1.0)
    wlen = np.where(nowrap, -1.0, radius * np.arccos(cosang))
    return wlen, t0, t1


def wrap(x0, x1, gpos, gmat, radius, is_cyl, side):
    """Spatial-tendon wrap over a sphere / cylinder (batched).  Returns
    (wlen[S] or -1, w0[S,3], w1[S,3]) in world coordinates."""
    S = x0.shape[0]
    p0 = np.einsum("sji,sj->si", gmat, x0 - gpos)
    p1 = np.einsum("sji,sj->si", gmat, x1 - gpos)
    close = (np.linalg.norm(p0, axis=-1) < MINVAL) | (np.linalg.norm(p1, axis=-1) < MINVAL)
    if is_cyl:
        ax0 = np.zeros((S, 3)); ax0[:, 0] = 1.0
        ax1 = np.zeros((S, 3)); ax1[:, 1] = 1.0
    else:
        ax0 = normalize(p0)
        nrm = np.cross(p0, p1)
        nn = np.linalg.norm(nrm, axis=-1)
        # degenerate (parallel) case: any vector orthogonal to ax0
        alt = np.zeros((S, 3))
        k = np.argmin(np.abs(ax0), axis=-1)
        alt[np.arange(S), k] = 1.0
        alt = np.cross(ax0, alt)
        nrm = np.where((nn < MINVAL)[:, None], alt, nrm)
        nrm = normalize(nrm)
        ax1 = normalize(np.cross(nrm, ax0))
    d0 = np.stack([np.sum(p0 * ax0, -1), np.sum(p0 * ax1, -1)], -1)
    d1 = np.stack([np.sum(p1 * ax0, -1), np.sum(p1 * ax1, -1)], -1)
    sd = None
    if side is not None:
        s = np.einsum("sji,sj->si", gmat, side - gpos)
        sd = normalize(np.stack([np.sum(s * ax0, -1), np.sum(s * ax1, -1)], -1))
    wlen, t0, t1 = wrap_circle(d0, d1, sd, radius)
    r0 = ax0 * t0[:, 0:1] + ax1 * t0[:, 1:2]
    r1 = ax0 * t1[:, 0:1] + ax1 * t1[:, 1:2]
    if is_cyl:
        L0 = np.sqrt((p0[:, 0] - t0[:, 0]) ** 2 + (p0[:, 1] - t0[:, 1]) ** 2)
        L1 = np.sqrt((p1[:, 0] - t1[:, 0]) ** 2 + (p1[:, 1] - t1[:, 1]) ** 2)
        tot = np.maximum(L0 + np.maximum(wlen, 0) + L1, MINVAL)
        r0[:, 2] = p0[:, 2] + (p1[:, 2] - p0[:, 2]) * L0 / tot
        r1[:, 2] = p0[:, 2] + (p1[:, 2] - p0[:, 2]) * (L0 + np.maximum(wlen, 0)) / tot
        h = np.abs(r1[:, 2] - r0[:, 2])
        wlen = np.where(wlen < 0, wlen, np.sqrt(wlen * wlen + h * h))
    wlen = np.where(close, -1.0, wlen)
    w0 = np.einsum("sij,sj->si", gmat, r0) + gpos
    w1 = np.einsum("sij,sj->si", gmat, r1) + gpos
    return wlen, w0, w1
