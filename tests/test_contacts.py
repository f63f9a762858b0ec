"""Contacts / equalities / free joint: oracle invariants (CPU) and HIP-vs-oracle parity (GPU).

The contact model restated in oracle/mmo_collision.inc is UNPINNED against MuJoCo (engine parity header); the CPU
tests here pin it against closed-form statics, the GPU tests pin the HIP general-row path against the oracle.
"""
import math

import numpy as np
import pytest

from myosuite_amd.model import synth
from myosuite_amd.model.spec import ModelSpec
from oracle import oracle as O


def _ball(mu=1.0):
    s = ModelSpec("ball", timestep=0.002)
    s.add_geom("floor", "world", "plane", (0, 0, 0))
    s.add_body("b", pos=(0, 0, 0.2), mass=1.0, inertia=(0.004, 0.004, 0.004))
    s.add_joint("root", "b", type="free")
    s.add_geom("s", "b", "sphere", (0.1,))
    s.add_contact_pair("floor", "s", condim=3, friction=(mu, 0.005, 0.0001))
    return s.compile()


def test_ball_rests_with_contact_force_equal_weight(oracle_lib):
    cm = _ball(); d = O.OracleData(O.OracleModel(cm))
    d.step(1000)
    assert d.ncon == 1 and d.nefc == 4            # one contact, four pyramid edges
    assert abs(d.efc_force[:4].sum() - 9.81) < 1e-4
    assert 0.099 < d.qpos[2] < 0.1               # sub-millimetre penetration of the soft contact
    assert np.abs(d.qvel).max() < 1e-8
    n = d.con_frame[0, :3]
    np.testing.assert_allclose(n, [0, 0, 1], atol=1e-12)


def test_axis_aligned_sliding_friction_is_mu_g(oracle_lib):
    for mu in (0.3, 0.8):
        s = ModelSpec("slider", timestep=0.002)
        s.add_geom("floor", "world", "plane", (0, 0, 0))
        s.add_body("b", pos=(0, 0, 0.1), mass=1.0, inertia=(0.004, 0.004, 0.004))
        for ax, v in (("x", (1, 0, 0)), ("y", (0, 1, 0)), ("z", (0, 0, 1))):
            s.add_joint(ax, "b", type="slide", axis=v)
        s.add_geom("c", "b", "capsule", (0.05, 0.1), quat=(math.cos(math.pi / 4), 0, math.sin(math.pi / 4), 0), pos=(0, 0, -0.05))
        s.add_contact_pair("c", "floor", condim=3, friction=(mu, 0.005, 0.0001))     # order is normalised by the compiler
        cm = s.compile()
        assert cm.arrays["GEOM_TYPE"][cm.arrays["PAIR_GEOM1"][0]] == 0
        d = O.OracleData(O.OracleModel(cm))
        d.step(500)
        assert d.ncon == 2 and d.nefc == 8        # plane-capsule: one contact per end cap
        d.qvel[0] = 1.0
        v = []
        for _ in range(31):
            d.step(); v.append(d.qvel[0])
        decel = (v[0] - v[30]) / (30 * 0.002)
        assert 0.85 * mu * 9.81 < decel <= 1.02 * mu * 9.81, (mu, decel)


def _free_geom_on_floor(gtype, size, quat=(1, 0, 0, 0), z=0.1, condim=3):
    s = ModelSpec("obj", timestep=0.002)
    s.add_geom("floor", "world", "plane", (0, 0, 0))
    s.add_body("b", pos=(0, 0, z), mass=1.0, inertia=(0.004, 0.004, 0.004), quat=quat)
    s.add_joint("root", "b", type="free")
    s.add_geom("g", "b", gtype, size)
    s.add_contact_pair("floor", "g", condim=condim, friction=(1.0, 0.005, 0.0001))
    return s.compile()


def test_contact_multiplicity_follows_mujocos_primitive_colliders(oracle_lib):
    """mmo_collision.inc header: plane-capsule 2 contacts with frames aligned with the capsule axis, plane-box up to 4 (corner
    order), plane-cylinder up to 4 (lowest rim point, the other cap, two triangle points), plane-ellipsoid 1 (support point),
    capsule-capsule 1 in general and 2 when the axes are parallel."""
    Y90 = (math.cos(math.pi / 4), 0, math.sin(math.pi / 4), 0)       # local z -> world x: a capsule / cylinder lying down
    # capsule lying on the plane: two contacts, tangent 1 of both frames along the capsule axis (mjc_PlaneCapsule)
    cm = _free_geom_on_floor("capsule", (0.05, 0.1), quat=Y90, z=0.0499)
    d = O.OracleData(O.OracleModel(cm)); d.forward()
    assert d.ncon == 2 and d.nefc == 8
    np.testing.assert_allclose(d.con_dist[:2], [-1e-4, -1e-4], atol=2e-8)
    np.testing.assert_allclose(d.con_pos[:2, 0], [0.1, -0.1], atol=2e-8)            # +axis end first
    for c in range(2):
        np.testing.assert_allclose(d.con_frame[c, :3], [0, 0, 1], atol=2e-8)
        np.testing.assert_allclose(np.abs(d.con_frame[c, 3:6]), [1, 0, 0], atol=2e-8)   # the capsule axis, not the default (0, 1, 0)
    # box flat on the plane: its four bottom corners, in corner order (x fastest)
    cm = _free_geom_on_floor("box", (0.1, 0.06, 0.03), z=0.0299)
    assert cm.npair == 2 and cm.arrays["PAIR_GEOM2"][0] == cm.arrays["PAIR_GEOM2"][1]    # a four-contact pair takes two entries
    d = O.OracleData(O.OracleModel(cm)); d.forward()
    assert d.ncon == 4 and d.nefc == 16
    np.testing.assert_allclose(d.con_pos[:4, :2], [[-0.1, -0.06], [0.1, -0.06], [-0.1, 0.06], [0.1, 0.06]], atol=2e-8)
    np.testing.assert_allclose(d.con_dist[:4], -1e-4, atol=2e-8)
    d.step(800)                                                                         # ... and it rests on them: sum of normal forces = weight
    assert d.ncon == 4 and abs(d.efc_force[:16].sum() - 9.81) < 1e-3 and np.abs(d.qvel).max() < 1e-6
    # box on an edge: only the two corners of that edge
    cm = _free_geom_on_floor("box", (0.1, 0.06, 0.03), quat=(math.cos(math.pi / 8), math.sin(math.pi / 8), 0, 0), z=0.0)
    d = O.OracleData(O.OracleModel(cm)); d.qpos[2] = 0.0636; d.forward()
    assert d.ncon == 2
    # cylinder standing on its cap: the lowest rim point (cylinder x axis when the cap is parallel to the plane), nothing on the
    # far cap, two triangle points at +-120 degrees
    cm = _free_geom_on_floor("cylinder", (0.05, 0.1), z=0.0999)
    d = O.OracleData(O.OracleModel(cm)); d.forward()
    assert d.ncon == 3
    r = np.hypot(d.con_pos[:3, 0], d.con_pos[:3, 1])
    np.testing.assert_allclose(r, 0.05, atol=2e-8)
    ang = np.sort(np.round(np.degrees(np.arctan2(d.con_pos[:3, 1], d.con_pos[:3, 0]))) % 360)
    np.testing.assert_allclose(ang, [0, 120, 240], atol=1e-6)
    # cylinder lying down: its line of contact is sampled at both caps
    cm = _free_geom_on_floor("cylinder", (0.05, 0.1), quat=Y90, z=0.0499)
    d = O.OracleData(O.OracleModel(cm)); d.forward()
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [-0.1, 0.1], atol=2e-8)
    # ellipsoid: one contact at the support point
    cm = _free_geom_on_floor("ellipsoid", (0.1, 0.06, 0.03), z=0.0299)
    d = O.OracleData(O.OracleModel(cm)); d.forward()
    assert d.ncon == 1 and abs(d.con_dist[0] + 1e-4) < 2e-8 and np.abs(d.con_pos[0, :2]).max() < 2e-8

    def two_capsules(quat2, pos2):
        s = ModelSpec("caps", timestep=0.002)
        s.add_body("a", pos=(0, 0, 0), mass=1.0, inertia=(0.004, 0.004, 0.004), quat=Y90)
        s.add_joint("ja", "a", type="free")
        s.add_geom("ca", "a", "capsule", (0.02, 0.1))
        s.add_body("b", pos=pos2, mass=1.0, inertia=(0.004, 0.004, 0.004), quat=quat2)
        s.add_joint("jb", "b", type="free")
        s.add_geom("cb", "b", "capsule", (0.02, 0.06))
        s.add_contact_pair("ca", "cb", condim=1)
        d_ = O.OracleData(O.OracleModel(s.compile())); d_.forward()
        return d_
    # parallel, overlapping: MuJoCo tries capsule 1's end caps first (+-0.1: 4 cm beyond capsule 2's ends, out of reach), then
    # capsule 2's end caps against segment 1: two contacts under the two ends of the SHORTER capsule
    d = two_capsules(Y90, (0.0, 0.0, 0.0399))
    assert d.ncon == 2
    np.testing.assert_allclose(d.con_pos[:2, 0], [0.06, -0.06], atol=2e-8)
    np.testing.assert_allclose(d.con_pos[:2, 2], 0.01995, atol=2e-8)
    np.testing.assert_allclose(d.con_dist[:2], -1e-4, atol=2e-8)
    # parallel, capsule 1 the shorter one in reach: its own end caps are the two contacts
    d = two_capsules(Y90, (0.07, 0.0, 0.0399))          # capsule 2 spans x in [0.01, 0.13]: capsule 1's +end (0.1) is under it, its -end is not
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [0.01, 0.1], atol=2e-8)
    # crossed: one contact at the closest points
    d = two_capsules((1, 0, 0, 0), (0.03, 0.0, 0.0399 + 0.06))                          # a vertical capsule standing on the lying one with its end cap
    assert d.ncon == 1
    np.testing.assert_allclose(d.con_pos[0], [0.03, 0.0, 0.01995], atol=2e-8)
    d = two_capsules((1, 0, 0, 0), (0.03, 0.0, 0.0401 + 0.06))                          # ... and 0.2 mm higher: out of reach (margin 0)
    assert d.ncon == 0
    d = two_capsules((math.cos(math.pi / 4), math.sin(math.pi / 4), 0, 0), (0.03, 0.0, 0.0399))   # local z -> world -y: crossed at right angles
    assert d.ncon == 1
    np.testing.assert_allclose(d.con_pos[0], [0.03, 0.0, 0.01995], atol=2e-8)
    np.testing.assert_allclose(d.con_dist[0], -1e-4, atol=2e-8)


def _capsule_over_box(quat, pos, margin=0.0, half=0.08, box=(0.05, 0.03, 0.02), radius=0.01):
    """a free capsule (radius, half length) posed over a box fixed in the world (top face at z = box[2]); condim 1"""
    s = ModelSpec("capbox", timestep=0.002)
    s.add_geom("box", "world", "box", box)
    s.add_body("c", pos=pos, mass=0.1, inertia=(1e-4, 1e-4, 1e-5), quat=quat)
    s.add_joint("jc", "c", type="free")
    s.add_geom("cap", "c", "capsule", (radius, half))
    s.add_contact_pair("cap", "box", condim=1, margin=margin)
    d = O.OracleData(O.OracleModel(s.compile())); d.forward()
    return d


def test_capsule_box_second_contact_and_convex_collider_contract(oracle_lib):
    """a4.3.  (i) mjc_CapsuleBox's result contract (mmo_collision.inc capsule_box_second): a capsule lying on a box
    face makes TWO contacts, one under each end -- clipped to the face when the capsule overhangs it --, a tilted capsule pressed
    into the face keeps the second one while its far sphere is within the margin, so does a capsule lying ALONG an edge; a capsule
    across a box edge, on a corner or standing on its end cap makes ONE.  (ii) the general convex collider's contract for capsule vs ellipsoid / cylinder: one
    contact, and the reported (distance, normal) satisfy the support-mapping optimality GJK terminates on -- the shape's support
    point along +n and the capsule's along -n are the witnesses, their gap along n is the distance -- to 1e-6."""
    Y90 = (math.cos(math.pi / 4), 0, math.sin(math.pi / 4), 0)       # capsule axis (local z) -> world x
    X90 = (math.cos(math.pi / 4), math.sin(math.pi / 4), 0, 0)       # capsule axis -> world -y
    r, top = 0.01, 0.02
    # lying on the top face, shorter than the face (half length 0.03 < 0.05): both end caps, each 1e-4 deep
    d = _capsule_over_box(Y90, (0.0, 0.0, top + r - 1e-4), half=0.03)
    assert d.ncon == 2 and d.nefc == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [-0.03, 0.03], atol=1e-9)
    np.testing.assert_allclose(d.con_dist[:2], -1e-4, atol=1e-9)
    np.testing.assert_allclose(d.con_frame[:2, :3], [[0, 0, -1], [0, 0, -1]], atol=1e-9)     # geom1 = capsule -> geom2 = box: downwards
    np.testing.assert_allclose(d.con_pos[:2, 2], top - 0.5e-4, atol=1e-9)
    # overhanging the face (half length 0.08 > 0.05): the two contacts sit at the ends of the FACE, not of the capsule
    d = _capsule_over_box(Y90, (0.0, 0.0, top + r - 1e-4), half=0.08)
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [-0.05, 0.05], atol=1e-9)
    # ... shifted: one end over the face, the other overhanging
    d = _capsule_over_box(Y90, (0.04, 0.0, top + r - 1e-4), half=0.03)
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [0.01, 0.05], atol=1e-9)
    # it rests there: two normal forces that add up to the weight, no rocking
    d = _capsule_over_box(Y90, (0.0, 0.0, top + r - 1e-4), half=0.03)
    d.step(600)
    assert d.ncon == 2 and abs(d.efc_force[:2].sum() - 0.1 * 9.81) < 1e-4 and abs(d.efc_force[0] - d.efc_force[1]) < 1e-6
    assert np.abs(d.qvel).max() < 1e-6
    # tilted by 2 mrad about y and pressed in: low end 1.6e-4 deep, high end 0.4e-4 deep -> still two contacts (MuJoCo's second sphere)
    th = 2e-3
    qt = (math.cos(math.pi / 4 + th / 2), 0, math.sin(math.pi / 4 + th / 2), 0)
    d = _capsule_over_box(qt, (0.0, 0.0, top + r - 1e-4), half=0.03)
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_dist[:2]), [-1e-4 - 0.03 * math.sin(th), -1e-4 + 0.03 * math.sin(th)], atol=1e-8)
    assert d.con_dist[0] < d.con_dist[1]                              # the closest sphere first
    # tilted by 20 mrad: the far sphere is 5e-4 above the face -> one contact; with a margin of 1e-3 it is a contact again
    th = 2e-2
    qt = (math.cos(math.pi / 4 + th / 2), 0, math.sin(math.pi / 4 + th / 2), 0)
    assert _capsule_over_box(qt, (0.0, 0.0, top + r - 1e-4), half=0.03).ncon == 1
    assert _capsule_over_box(qt, (0.0, 0.0, top + r - 1e-4), half=0.03, margin=1e-3).ncon == 2
    # across the box's top edge x = +0.05 at 45 degrees (axis in the x-z plane, pointing down-outwards): nearest feature is the edge -> one
    q45 = (math.cos(math.pi / 8), 0, math.sin(math.pi / 8), 0)
    c = np.array([0.05, 0.0, top]) + (r - 1e-4) * np.array([math.sin(math.pi / 4), 0, math.cos(math.pi / 4)])   # axis point nearest the edge
    d = _capsule_over_box((math.cos(3 * math.pi / 8), 0, math.sin(3 * math.pi / 8), 0), tuple(c), half=0.03)
    assert d.ncon == 1 and abs(d.con_dist[0] + 1e-4) < 1e-8
    np.testing.assert_allclose(np.abs(d.con_frame[0, :3]), [math.sin(math.pi / 4), 0, math.cos(math.pi / 4)], atol=1e-7)
    # lying ALONG the top edge y = +0.03 (axis -> x), outside it on the diagonal: the stretch beside the edge, two contacts on the edge line
    e45 = np.array([0.0, math.sin(math.pi / 4), math.cos(math.pi / 4)])
    d = _capsule_over_box(Y90, tuple(np.array([0.0, 0.03, top]) + (r - 1e-4) * e45), half=0.03)
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [-0.03, 0.03], atol=1e-9)
    np.testing.assert_allclose(d.con_dist[:2], -1e-4, atol=1e-8)
    np.testing.assert_allclose(np.abs(d.con_frame[:2, :3]), [np.abs(e45)] * 2, atol=1e-7)
    d = _capsule_over_box(Y90, tuple(np.array([0.0, 0.03, top]) + (r - 1e-4) * e45), half=0.08)       # longer than the edge: clipped to it
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 0]), [-0.05, 0.05], atol=1e-9)
    # lying ALONG the edge direction but over the face interior, crossing the y edges (axis -> y, half 0.08 > 0.03): clipped to the face in y
    d = _capsule_over_box(X90, (0.0, 0.0, top + r - 1e-4), half=0.08)
    assert d.ncon == 2
    np.testing.assert_allclose(np.sort(d.con_pos[:2, 1]), [-0.03, 0.03], atol=1e-9)
    # standing on its end cap: one contact (the second sphere, at the top end, is far out of reach)
    d = _capsule_over_box((1, 0, 0, 0), (0.01, 0.0, top + 0.03 + r - 1e-4), half=0.03)
    assert d.ncon == 1 and abs(d.con_dist[0] + 1e-4) < 1e-8
    # over a corner: one
    cdir = np.array([1.0, 1.0, 1.0]) / math.sqrt(3)
    d = _capsule_over_box(Y90, tuple(np.array([0.05, 0.03, top]) + (r - 1e-4) * cdir + np.array([0.03, 0, 0])), half=0.03)
    assert d.ncon == 1 and abs(d.con_dist[0] + 1e-4) < 1e-8

    # a sphere against the three convex primitives = the capsule of zero length: one contact at the closest point
    for gname, size, zc in (("box", (0.05, 0.03, 0.02), 0.02), ("cylinder", (0.03, 0.02), 0.02), ("ellipsoid", (0.05, 0.03, 0.02), 0.02)):
        s_ = ModelSpec("sphcvx", timestep=0.002)
        s_.add_geom("shape", "world", gname, size)
        off = (0.0, 0.0) if gname == "ellipsoid" else (0.004, -0.003)
        s_.add_body("b", pos=(off[0], off[1], zc + 0.01 - 1e-4), mass=0.1, inertia=(1e-5, 1e-5, 1e-5)); s_.add_joint("jb", "b", type="free")
        s_.add_geom("ball", "b", "sphere", (0.01,))
        s_.add_contact_pair("ball", "shape", condim=1)
        cm_ = s_.compile()
        d = O.OracleData(O.OracleModel(cm_)); d.forward()
        assert d.ncon == 1 and d.warn == 0, gname
        assert abs(d.con_dist[0] + 1e-4) < 1e-8
        np.testing.assert_allclose(d.con_frame[0, :3], [0, 0, -1], atol=1e-7)
    # ---- (ii) capsule vs ellipsoid / cylinder: the convex collider's result contract
    def support_shape(gtype, size, R, x, n):
        nl = R.T @ n
        if gtype == 4:
            loc = size * size * nl / math.sqrt(((size * nl) ** 2).sum())
        else:                                                        # cylinder (radius, half length)
            rho = math.hypot(nl[0], nl[1])
            loc = np.array([size[0] * nl[0] / rho if rho > 1e-12 else 0.0, size[0] * nl[1] / rho if rho > 1e-12 else 0.0,
                            size[1] * (1 if nl[2] >= 0 else -1)])
        return x + R @ loc
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(60):
        gname, gtype = (("ellipsoid", 4), ("cylinder", 5))[trial % 2]
        size = rng.uniform(0.015, 0.045, 3) if gtype == 4 else np.array([rng.uniform(0.015, 0.03), rng.uniform(0.02, 0.045), 0.0])
        qs = rng.standard_normal(4); qs /= np.linalg.norm(qs)
        qc = rng.standard_normal(4); qc /= np.linalg.norm(qc)
        dirn = rng.standard_normal(3); dirn /= np.linalg.norm(dirn)
        s = ModelSpec("capcvx", timestep=0.002)
        s.add_body("s", pos=(0, 0, 0), mass=1.0, inertia=(1e-3, 1e-3, 1e-3), quat=tuple(qs)); s.add_joint("js", "s", type="free")
        s.add_geom("shape", "s", gname, tuple(size))
        s.add_body("c", pos=tuple(rng.uniform(0.03, 0.075) * dirn), mass=0.1, inertia=(1e-4, 1e-4, 1e-5), quat=tuple(qc)); s.add_joint("jc", "c", type="free")
        s.add_geom("cap", "c", "capsule", (0.008, 0.03))
        s.add_contact_pair("cap", "shape", condim=1, margin=0.05)
        cm = s.compile()
        d = O.OracleData(O.OracleModel(cm)); d.forward()
        assert d.ncon <= 1
        if d.ncon == 0:
            continue
        g_cap, g_shape = cm.names["geom"]["cap"], cm.names["geom"]["shape"]
        n = d.con_frame[0, :3].copy()                                 # from the capsule (geom1) to the shape (geom2)
        xs, Rs = d.geom_xpos[g_shape], d.geom_xmat[g_shape].reshape(3, 3)
        xc, uc = d.geom_xpos[g_cap], d.geom_xmat[g_cap].reshape(3, 3)[:, 2]
        w_shape = support_shape(gtype, size, Rs, xs, -n)              # the shape's extreme point towards the capsule
        w_cap = xc + 0.03 * uc * (1 if uc @ n >= 0 else -1) + 0.008 * n   # the capsule's extreme point towards the shape
        gap = (w_shape - w_cap) @ n                                   # separation of the two support planes along n
        if d.con_dist[0] > 1e-5:                                      # separated: the support planes' gap IS the distance (GJK's exit test)
            assert abs(gap - d.con_dist[0]) < 1e-6, (gname, trial, gap, d.con_dist[0])
        else:                                                         # touching / penetrating: depth along n, same witnesses
            assert gap <= d.con_dist[0] + 1e-6
        # the contact point lies halfway between the two surfaces on the line of the normal
        # (unique witnesses only: an ellipsoid against a capsule whose axis is not perpendicular to the normal)
        mid_err = np.linalg.norm(np.cross(d.con_pos[0] - 0.5 * (w_shape + w_cap), n)) if (d.con_dist[0] > 1e-5 and gtype == 4) else 0.0
        assert mid_err < 1e-5 or abs(uc @ n) < 1e-3
        checked += 1
    assert checked >= 20


def test_leg_model_dimensions_and_names(oracle_lib):
    cm = synth.get_model("leg")
    # SURVEY 8d / walk_v0.py: nq 35, nv 34, 80 muscles; obs = 33+34+2+4+2+1+6+1+3*80+80 = 403
    assert (cm.nq, cm.nv, cm.nu, cm.na) == (35, 34, 80, 80)
    assert cm.neq == 14 and cm.npair == 8 and cm.njmax <= 64
    for n in ("hip_flexion_l", "hip_flexion_r", "hip_adduction_l", "hip_adduction_r", "hip_rotation_l", "hip_rotation_r",
              "knee_angle_r", "ankle_angle_l"):
        assert n in cm.names["joint"]
    for b in ("pelvis", "torso", "talus_l", "talus_r"):
        assert b in cm.names["body"]
    assert cm.key_qpos.shape == (4, 35) and cm.key_qvel.shape == (4, 34)
    d = O.OracleData(O.OracleModel(cm))
    d.qpos[:] = cm.key_qpos[0]; d.qpos[2] -= 1e-6; d.forward()
    assert d.ncon >= 2 and np.abs(d.con_dist[:d.ncon]).max() < 2e-3   # grounded keyframe: the lowest foot spheres touch the floor
    # independent coordinates of the keyframes are the reference's (myolegs_chasetag.xml:53-56)
    assert abs(cm.key_qpos[2][7 + 5] - 1.227) < 1e-12 and abs(cm.key_qvel[2][1] + 1.5) < 1e-12 and abs(cm.key_qvel[2][6] - 4.9066) < 1e-12
    assert np.abs(d.efc_pos[:14]).max() < 1e-8                       # knee couplings satisfied by the keyframes (float32 polycoef)
    d.qpos[:] = cm.key_qpos[2]; d.forward()
    assert np.abs(d.efc_pos[:14]).max() < 1e-6


def test_leg_free_joint_mass_matrix_matches_numpy(oracle_lib):
    from myosuite_amd.model import kin_np as K
    cm = synth.get_model("leg")
    km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
    d = O.OracleData(O.OracleModel(cm))
    rng = np.random.default_rng(0)
    q = cm.key_qpos[2].copy(); q[7:] += rng.uniform(-0.3, 0.3, 28)
    qq = rng.standard_normal(4); q[3:7] = qq / np.linalg.norm(qq)
    d.qpos[:] = q; d.forward()
    M = km.mass_matrix(q[None])[0] + np.diag(cm.arrays["DOF_ARMATURE"].astype(float))
    assert np.abs(d.full_M() - M).max() < 1e-12
    assert np.abs(km.tendon_length(q[None])[0] - d.ten_length).max() < 1e-12
    # total momentum check of the free-floating tree: gravity only accelerates the root translation
    d.qvel[:] = 0; d.act[:] = 0; d.ctrl[:] = 0
    d.qpos[2] += 1.0; d.forward()                                   # lifted: no contacts
    assert d.ncon == 0


# ----------------------------------------------------------------------------------- GPU
def _states(cm, name, n, rng):
    q = np.tile(cm.qpos0.astype(np.float64), (n, 1))
    if name == "plane_toy":
        # three free bodies at random tilts near the plane (faces / edges / corners / rims down), the bar over, on and beside the rail
        for k, (h, spread) in enumerate(((0.05, 0.03), (0.07, 0.04), (0.045, 0.03))):
            o = 7 * k
            q[:, o + 2] = h + rng.uniform(-spread, spread, n) - 0.06 * q[:, o]      # follow the 3.4 deg tilt (z falls with +x)
            kind = rng.integers(0, 3, n)
            qq = rng.standard_normal((n, 4)) * np.where(kind == 0, 0.02, np.where(kind == 1, 0.15, 1.0))[:, None] + np.array([1, 0, 0, 0])
            q[:, o + 3:o + 7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        q[:, 21] = rng.uniform(-0.004, 0.003, n); q[:, 22] = rng.uniform(-0.12, 0.12, n)
        # the rod over the anvil (top face z = 0.12, |x| <= 0.06, |y| <= 0.04 about (0, -0.5)): lying on the face (two contacts), tilted by
        # a few mrad (two while the far sphere still penetrates, else one), at a random attitude (end cap / across an edge: one), or clear
        kind = rng.integers(0, 4, n)
        tilt = np.where(kind == 0, 0.0, np.where(kind == 1, rng.uniform(-6e-3, 6e-3, n), rng.uniform(-0.5, 0.5, n)))
        yaw = rng.uniform(-0.6, 0.6, n)
        for e in range(n):
            cy, sy = math.cos(math.pi / 4 + tilt[e] / 2), math.sin(math.pi / 4 + tilt[e] / 2)
            qy = np.array([cy, 0.0, sy, 0.0]); qz = np.array([math.cos(yaw[e] / 2), 0.0, 0.0, math.sin(yaw[e] / 2)])
            w1, x1, y1, z1 = qz; w2, x2, y2, z2 = qy                                   # yaw about world z, then the lying-down rotation
            q[e, 26:30] = [w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                           w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2]
        q[:, 23] = rng.uniform(-0.05, 0.05, n); q[:, 24] = -0.5 + rng.uniform(-0.03, 0.03, n)
        q[:, 25] = 0.135 + np.where(kind == 3, 0.01, rng.uniform(-4e-4, -5e-5, n)) + 0.05 * np.abs(np.sin(tilt)) * (kind == 2)
        # every other "flat" env lies ALONG the anvil's top edge y = -0.46 instead (no yaw, pressed in 1e-4 on the edge's diagonal):
        # mjc_CapsuleBox's two contacts beside an edge
        along = np.nonzero(kind == 0)[0][1::2]
        q[along, 26:30] = [math.cos(math.pi / 4), 0.0, math.sin(math.pi / 4), 0.0]
        q[along, 24] = -0.46 + (0.015 - 1e-4) * math.sin(math.pi / 4); q[along, 25] = 0.12 + (0.015 - 1e-4) * math.cos(math.pi / 4)
        v = rng.standard_normal((n, cm.nv)) * 0.3
    elif name == "contact_toy":
        q[:, 2] += rng.uniform(-0.04, 0.05, n)
        qq = rng.standard_normal((n, 4)) * 0.2 + np.array([1, 0, 0, 0]); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        q[:, 7:10] += rng.uniform(-0.05, 0.05, (n, 3)); q[:, 9] -= rng.uniform(0.0, 0.2, n)
        q[:, 10] = rng.uniform(-1.3, 1.3, n); q[:, 11] = rng.uniform(-0.45, 0.12, n); q[:, 12] = rng.uniform(-0.1, 0.1, n)
        v = rng.standard_normal((n, cm.nv)) * 0.5
    else:
        for e in range(n):
            q[e] = cm.key_qpos[(0, 2, 3)[e % 3]]
        q[:, 7:] += rng.uniform(-0.15, 0.15, (n, cm.nq - 7))
        q[:, 2] += rng.uniform(-0.03, 0.02, n)
        qq = rng.standard_normal((n, 4)) * 0.05 + np.array([1, 0, 0, 0]); q[:, 3:7] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
        v = rng.standard_normal((n, cm.nv)) * 0.3
    return q.astype(np.float32), v.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["contact_toy", "leg", "plane_toy"])
def test_gpu_general_rows_forward_and_rollout_match_oracle(oracle_lib, name):
    import torch
    from myosuite_amd import engine as E
    cm = synth.get_model(name); hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(1)
    n = 24
    q, v = _states(cm, name, n, rng)
    act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.random((n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v)); st.act.copy_(torch.from_numpy(act))
    dump = E.debug_dump(hm, st, torch.from_numpy(ctrl).cuda()).cpu().numpy()
    ds = []
    saw_contact = saw_multi_iter = 0
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.act[:] = act[e]; d.ctrl[:] = ctrl[e]
        d.forward(); ds.append(d)
        saw_contact += d.ncon > 0; saw_multi_iter += d.solver_niter > 1
        for nm, ref in (("qaccsm", d.qacc_smooth), ("qacc", d.qacc), ("smooth", d.qfrc_smooth)):
            got = dump[e, hm.layout(nm):hm.layout(nm) + ref.size]
            assert np.abs(got - ref).max() < 2e-4 * max(1e-9, np.abs(ref).max()), (nm, e)
        got = dump[e, hm.layout("qfrccon"):hm.layout("qfrccon") + cm.nv]
        assert np.abs(got - d.qfrc_constraint).max() < 2e-4 * max(1.0, np.abs(d.qfrc_smooth).max())
        assert int(dump[e, hm.layout("scal")]) == d.solver_niter          # same Newton path
    assert saw_contact >= n // 2 and saw_multi_iter >= 1
    if name == "plane_toy":      # every multiplicity of the multi-contact colliders occurs in the sample
        per_pair = np.zeros((n, cm.npair), int)
        for e, d in enumerate(ds):
            for c_ in range(d.ncon):
                per_pair[e, d.con_pair[c_]] += 1
        box, cyl, ell, cap = per_pair[:, 0] + per_pair[:, 1], per_pair[:, 2] + per_pair[:, 3], per_pair[:, 4], per_pair[:, 5]
        assert box.max() == 4 and {1, 2} <= set(box) and cyl.max() >= 3 and ell.max() == 1 and cap.max() == 2, (set(box), set(cyl), set(cap))
        rod = per_pair[:, 6]             # mjc_CapsuleBox: none / the closest sphere / both spheres along the face
        assert {0, 1, 2} <= set(rod) and rod.max() == 2, set(rod)
    c = torch.from_numpy(ctrl).cuda()
    for _ in range(4):
        E.step(hm, st, c, 25)
        for d in ds:
            d.step(25)
    qo = np.array([d.qpos for d in ds])
    err = np.abs(st.qpos.cpu().numpy() - qo).max(axis=1)
    assert int(st.status.cpu().max()) == 0 and max(d.warn for d in ds) == 0
    worst = int(err.argmax())
    by_dof = np.abs(st.qpos.cpu().numpy()[worst] - qo[worst])
    print("ROLLOUT-ERR", name, "worst env", worst, "by qpos:", " ".join(f"{x:.1e}" for x in by_dof), "| q0:", " ".join(f"{x:.5f}" for x in q[worst]))
    assert np.median(err) < 2e-5 and err.max() < 1e-3, (np.median(err), err.max(), worst, np.round(by_dof, 6).tolist(), q[worst].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["contact_toy", "plane_toy"])
def test_gpu_condim4_torsional_rows_match_the_oracle(oracle_lib, name):
    """condim 4 (torsional friction: the reference's pen, myohand_pen.xml:34): every frictional pair of the two collider scenes
    switched to condim 4 with a visible torsional coefficient -- six pyramid rows per contact.  Row counts, constrained accelerations
    and constraint forces of one forward pass, then 100 free-running substeps, HIP vs oracle; and the torsional pair matters (the same
    states with condim 3 give a different constrained acceleration)."""
    import torch
    from myosuite_amd import engine as E
    spec = synth.builders()[name]()
    for p in spec.pairs:
        if p["condim"] == 3:
            p["condim"] = 4; p["friction"] = (p["friction"][0], 0.02, p["friction"][2])
    if name == "plane_toy":          # 14 possible contacts x 6 rows do not fit one row per lane: explicit bounds (surplus dropped whole, flagged)
        spec.nconmax, spec.njmax = 10, 60
    cm = spec.compile(); cm3 = synth.get_model(name)
    hm = E.HipModel(cm); om = O.OracleModel(cm); om3 = O.OracleModel(cm3)
    rng = np.random.default_rng(3)
    n = 32
    q, v = _states(cm3, name, n, rng)
    v[:, 3:6] += rng.standard_normal((n, 3)).astype(np.float32) * 3.0            # spin the first free body: torsion has something to resist
    act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.random((n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
    if cm.na:
        st.act.copy_(torch.from_numpy(act))
    dv = E.Derived(hm, n, ["qacc", "nefc"])
    c = torch.from_numpy(ctrl).cuda()
    E.forward(hm, st, c, dv); torch.cuda.synchronize()
    ga, gn = dv["qacc"].cpu().numpy().astype(np.float64), dv["nefc"].cpu().numpy()
    ds, six, differs = [], 0, 0
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.ctrl[:] = ctrl[e]
        d3 = O.OracleData(om3); d3.qpos[:] = q[e]; d3.qvel[:] = v[e]; d3.ctrl[:] = ctrl[e]
        if cm.na:
            d.act[:] = act[e]; d3.act[:] = act[e]
        d.forward(); d3.forward(); ds.append(d)
        assert d.nefc == gn[e], (e, d.nefc, gn[e])
        assert d.nefc >= d3.nefc
        six += d.nefc > d3.nefc
        differs += np.abs(d.qacc - d3.qacc).max() > 1e-3 * max(1.0, np.abs(d.qacc).max())
        assert np.abs(ga[e] - d.qacc).max() < 3e-4 * max(1.0, np.abs(d.qacc).max()), (e, np.abs(ga[e] - d.qacc).max(), np.abs(d.qacc).max())
    assert six >= n // 2 and differs >= 3, (six, differs)
    for _ in range(4):
        E.step(hm, st, c, 25)
        for d in ds:
            d.step(25)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert int((st.status.cpu() & ~8).max()) == 0 and max(d.warn & ~6 for d in ds) == 0      # (2 | 4: njmax / nconmax overflow under the explicit bounds)
    assert np.median(err) < 5e-5 and np.quantile(err, 0.9) < 2e-3, (np.median(err), err.max())


@pytest.mark.gpu
def test_gpu_capsule_box_narrow_phase_matches_oracle_on_adversarial_poses(oracle_lib):
    """2048 rod poses over the anvil of `plane_toy`, biased to the places where the capsule-box rule branches: exactly flat / tilted
    by micro- to milliradians on the face, overhanging an edge and tipping over it, lying along an edge, crossing edges and corners,
    penetrating by 1e-6...1e-3, clear by as little -- one forward pass of the whole batch, and for EVERY env the number of
    constraint rows (4 per rod contact; the other bodies are parked in the air) and the constrained acceleration against the oracle
    on the same state.  Envs whose row count differs must be rare (a sphere within fp32 rounding of its margin) and are excluded
    from the error statistics."""
    import torch
    from myosuite_amd import engine as E
    cm = synth.get_model("plane_toy"); hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(11)
    n = 2048
    q = np.tile(cm.qpos0.astype(np.float64), (n, 1))
    for k in range(3):
        q[:, 7 * k + 2] += 1.0                               # box / cylinder / ellipsoid: a metre above the plane
    q[:, 21] = 0.2                                           # the bar: well above the rail
    kind = rng.integers(0, 6, n)
    r, top, hx, hy, hl = 0.015, 0.12, 0.06, 0.04, 0.05       # rod radius, anvil top, anvil half sizes, rod half length
    pen = -np.exp(rng.uniform(np.log(1e-6), np.log(1e-3), n)) * np.where(rng.random(n) < 0.25, -1.0, 1.0)     # signed gap of the closest sphere
    tilt = np.where(kind == 0, 0.0, np.where(kind == 1, np.exp(rng.uniform(np.log(1e-6), np.log(2e-2), n)) * rng.choice([-1, 1], n),
                    np.where(kind == 5, rng.uniform(-0.8, 0.8, n), rng.uniform(-3e-3, 3e-3, n))))
    yaw = np.where(kind == 3, 0.0, np.where(kind == 4, rng.uniform(0.3, 1.2, n), rng.uniform(-0.5, 0.5, n)))
    x = np.where(kind == 2, rng.uniform(0.02, 0.09, n), rng.uniform(-0.03, 0.03, n))        # kind 2: one end overhangs the +x edge
    y = rng.uniform(-0.02, 0.02, n)
    z = top + r + pen + hl * np.abs(np.sin(tilt))            # the lower end's sphere sits `pen` from the face (face poses)
    diag = math.sqrt(0.5)
    along = kind == 3                                        # along the +y top edge, on its diagonal
    y = np.where(along, hy + (r + pen) * diag, y); z = np.where(along, top + (r + pen) * diag, z)
    tilt = np.where(along, 0.0, tilt)
    for e in range(n):
        cy, sy = math.cos(math.pi / 4 + tilt[e] / 2), math.sin(math.pi / 4 + tilt[e] / 2)
        w1, z1 = math.cos(yaw[e] / 2), math.sin(yaw[e] / 2)
        q[e, 26:30] = [w1 * cy, -z1 * sy, w1 * sy, z1 * cy]                                  # yaw about world z after the lying-down rotation
    # kind 4: ACROSS the +y top edge (tangent to it at an angle), every third one at the (+x, +y) corner instead: the rod lies in the
    # plane at distance r + pen from a supporting plane of the box through the edge / corner point, so that point is its closest one
    def quat_from_z(u):
        c = float(u[2]); ax = np.array([-u[1], u[0], 0.0]); sn = np.linalg.norm(ax)
        if sn < 1e-12:
            return np.array([1.0, 0, 0, 0]) if c > 0 else np.array([0.0, 1, 0, 0])
        half = 0.5 * math.atan2(sn, c)
        return np.concatenate([[math.cos(half)], math.sin(half) * ax / sn])
    for e in np.nonzero(kind == 4)[0]:
        corner = e % 3 == 0
        if corner:
            nrm = rng.uniform(0.25, 1.0, 3); cp = np.array([hx, hy, top])
        else:
            th = rng.uniform(0.3, 1.25); nrm = np.array([0.0, math.sin(th), math.cos(th)]); cp = np.array([rng.uniform(-0.03, 0.03), hy, top])
        nrm /= np.linalg.norm(nrm)
        t1_ = np.cross(nrm, [1.0, 0.0, 0.0] if not corner else rng.standard_normal(3)); t1_ /= np.linalg.norm(t1_)
        t2_ = np.cross(nrm, t1_)
        psi = rng.uniform(0.3, 1.2) if not corner else rng.uniform(0, 2 * math.pi)
        u_ = math.cos(psi) * t2_ + math.sin(psi) * t1_            # (edge: t2_ = +-x, so psi is the crossing angle)
        c_ = cp + (r + pen[e]) * nrm + rng.uniform(-0.6, 0.6) * hl * u_
        x[e], y[e], z[e] = c_
        q[e, 26:30] = quat_from_z(u_)
    q[:, 23] = x; q[:, 24] = -0.5 + y; q[:, 25] = z
    q32 = q.astype(np.float32)
    v = (rng.standard_normal((n, cm.nv)) * 0.2).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q32)); st.qvel.copy_(torch.from_numpy(v))
    dv = E.Derived(hm, n, ["qacc", "nefc"])
    ctrl = torch.zeros(n, max(cm.nu, 1), device="cuda")[:, :cm.nu].contiguous()
    E.forward(hm, st, ctrl, dv)
    torch.cuda.synchronize()
    gn, ga = dv["nefc"].cpu().numpy(), dv["qacc"].cpu().numpy().astype(np.float64)
    d = O.OracleData(om)
    on = np.zeros(n, int); rel = np.zeros(n)
    for e in range(n):
        d.qpos[:] = q32[e]; d.qvel[:] = v[e]; d.qacc_warmstart[:] = 0; d.forward()
        on[e] = d.nefc
        rel[e] = np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max())
    mism = on != gn
    hist = {int(k_): int((on == k_).sum()) for k_ in np.unique(on)}
    print(f"capsule-box sweep: oracle row counts {hist}; row-count mismatches {int(mism.sum())} of {n} (by kind {[int(mism[kind == k_].sum()) for k_ in range(6)]}); "
          f"rel |dqacc| median {np.median(rel[~mism]):.1e} max {rel[~mism].max():.1e}")
    assert set(hist) == {0, 4, 8} and min(hist.values()) >= n // 20, hist                    # none / one / two rod contacts all well represented
    assert mism.sum() <= n // 200, (int(mism.sum()), np.nonzero(mism)[0][:10], on[mism][:10], gn[mism][:10])
    assert rel[~mism].max() < 2e-3 and np.quantile(rel[~mism], 0.99) < 3e-4, (rel[~mism].max(), np.quantile(rel[~mism], 0.99))
    assert int(st.status.cpu().max()) & ~1 == 0


@pytest.mark.gpu
def test_gpu_sphere_vs_convex_primitives_matches_oracle(oracle_lib):
    """sphere vs box / cylinder / ellipsoid (the zero-length capsule of the capsule-convex collider): 256 random ball positions around
    each of three world-fixed shapes -- faces, edges, corners, caps, rims; touching, penetrating, just clear -- one forward pass,
    row counts and constrained acceleration of every env against the oracle."""
    import torch
    from myosuite_amd import engine as E
    s = ModelSpec("sphere_convex_toy", timestep=0.002)
    shapes = (("box", (0.05, 0.03, 0.02)), ("cylinder", (0.03, 0.025)), ("ellipsoid", (0.05, 0.03, 0.02)))
    for k, (gname, size) in enumerate(shapes):
        s.add_geom(f"shape{k}", "world", gname, size, pos=(0.3 * k, 0.0, 0.0), quat=(0.9, 0.1 * k, -0.2, 0.3))
    for k in range(3):
        s.add_body(f"b{k}", pos=(0.3 * k, 0.0, 0.08), mass=0.1, inertia=(1e-5, 1e-5, 1e-5)); s.add_joint(f"j{k}", f"b{k}", type="free")
        s.add_geom(f"ball{k}", f"b{k}", "sphere", (0.012,))
        s.add_contact_pair(f"ball{k}", f"shape{k}", condim=3, margin=0.001)
    cm = s.compile(); hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(4)
    n = 256
    q = np.tile(cm.qpos0.astype(np.float64), (n, 1))
    for k, (gname, size) in enumerate(shapes):
        # a point on / near the shape's surface in its own frame, pushed out along a random direction by r + gap
        dirs = rng.standard_normal((n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        ext = np.array(size if len(size) == 3 else (size[0], size[0], size[1]))
        surf = dirs * ext * rng.uniform(0.9, 1.4, (n, 1))
        gap = np.where(rng.random(n) < 0.6, 1.0, -1.0) * np.exp(rng.uniform(np.log(1e-5), np.log(1e-2), n))
        d0 = O.OracleData(om); d0.forward()
        R = d0.geom_xmat[cm.names["geom"][f"shape{k}"]].reshape(3, 3); xs = d0.geom_xpos[cm.names["geom"][f"shape{k}"]]
        # project to the surface by the oracle's own signed distance (test hook), then offset
        gt = {"box": 6, "cylinder": 5, "ellipsoid": 4}[gname]
        for e in range(n):
            p = surf[e].copy()
            for _ in range(3):
                sd, _, g = O.seg_shape(gt, ext if gt != 5 else np.array([size[0], size[1], 0.0]), p, np.array([0.0, 0.0, 1.0]), 0.0)
                p = p - sd * g
            sd, _, g = O.seg_shape(gt, ext if gt != 5 else np.array([size[0], size[1], 0.0]), p, np.array([0.0, 0.0, 1.0]), 0.0)
            c = p + (0.012 + gap[e]) * g
            q[e, 7 * k:7 * k + 3] = xs + R @ c
    q32 = q.astype(np.float32); v = (0.2 * rng.standard_normal((n, cm.nv))).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q32)); st.qvel.copy_(torch.from_numpy(v))
    dv = E.Derived(hm, n, ["qacc", "nefc"])
    E.forward(hm, st, None, dv)
    torch.cuda.synchronize()
    gn, ga = dv["nefc"].cpu().numpy(), dv["qacc"].cpu().numpy().astype(np.float64)
    d = O.OracleData(om); on = np.zeros(n, int); rel = np.zeros(n)
    for e in range(n):
        d.qpos[:] = q32[e]; d.qvel[:] = v[e]; d.qacc_warmstart[:] = 0; d.forward()
        on[e] = d.nefc; rel[e] = np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max())
        assert d.warn == 0
    mism = on != gn
    print(f"sphere-convex sweep: oracle row counts {({int(k_): int((on == k_).sum()) for k_ in np.unique(on)})}, mismatches {int(mism.sum())}, rel |dqacc| max {rel[~mism].max():.1e}")
    assert {4, 8, 12} <= set(on.tolist()) and (on < 12).sum() >= n // 4 and mism.sum() <= 2 and rel[~mism].max() < 2e-3
    assert int(st.status.cpu().max()) == 0


def test_friction_loss_rows_oracle(oracle_lib):
    """Dry joint friction (MuJoCo `frictionloss`): the row force saturates at +-frictionloss, holds a load below the bound
    (up to the soft-constraint creep) and dissipates energy."""
    from myosuite_amd.model.spec import ModelSpec

    def toy(f):
        s = ModelSpec("fric_pendulum")
        s.add_body("arm", "world", pos=(0, 0, 1), mass=1.0, ipos=(0.1, 0, 0), inertia=(1e-3, 1e-3, 1e-3))
        s.add_joint("hinge", "arm", "hinge", axis=(0, 1, 0), frictionloss=f)
        return s.compile()
    tau_g = 1.0 * 9.81 * 0.1
    for f in (0.5, 1.2):
        cm = toy(f)
        assert cm.njmax == 1
        d = O.OracleData(O.OracleModel(cm)); d.reset(); d.forward()
        assert d.nefc == 1
        if f < tau_g:      # saturated: force = -f exactly
            assert abs(d.qfrc_constraint[0] + f) < 1e-12
            assert abs(d.qacc[0] - (tau_g - f) / (1e-3 + 0.01)) < 1e-4      # model constants are stored as f32
        else:              # below the bound: held, up to the impedance (d0 = 0.9 -> at least 85 % of the load)
            assert -tau_g < d.qfrc_constraint[0] < -0.85 * tau_g
            d.step(500)
            assert abs(d.qpos[0]) < 0.12 and abs(d.qvel[0]) < 0.12
    # free swing with friction loses energy monotonically (sampled at velocity zero crossings via |q| peaks)
    cm = toy(0.2)
    d = O.OracleData(O.OracleModel(cm)); d.reset(); d.qpos[0] = -1.0
    peaks, prev_v = [], 0.0
    for k in range(4000):
        d.step(1)
        if prev_v * d.qvel[0] < 0:
            peaks.append(abs(d.qpos[0] - np.pi / 2))
        prev_v = d.qvel[0]
    assert len(peaks) >= 3 and all(b < a for a, b in zip(peaks, peaks[1:]))


@pytest.mark.gpu
def test_friction_loss_gpu_matches_oracle(oracle_lib):
    """friction_toy (three friction-loss dofs with different solref/solimp, limits, an equality, motors): teacher-forced
    per-substep parity of the fused kernel against the oracle, and a short free run."""
    import torch
    from myosuite_amd import engine as E
    cm = synth.get_model("friction_toy")
    assert cm.njmax == 1 + 3 + 2
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    n = 16
    rng = np.random.default_rng(4)
    q = rng.uniform(-0.5, 0.8, (n, cm.nq)); q[:, 3] = 0.05 * q[:, 1]
    v = rng.standard_normal((n, cm.nv)) * np.array([1.0, 1.0, 3.0, 0.05])
    v[: n // 4] = 0.0                                            # some envs start at rest: stick regime
    st = E.BatchState(hm, n)
    ds = [O.OracleData(om) for _ in range(n)]
    worst = 0.0
    for k in range(60):
        ctrl = rng.uniform(-1, 1, (n, cm.nu)).astype(np.float32)
        if k % 3 == 0:
            ctrl[:, :] *= 0.05                                   # small torques: below the friction bounds
        st.qpos.copy_(torch.from_numpy(q.astype(np.float32))); st.qvel.copy_(torch.from_numpy(v.astype(np.float32)))
        E.step(hm, st, torch.from_numpy(ctrl).cuda(), 1)
        qg, vg = st.qpos.cpu().numpy().astype(np.float64), st.qvel.cpu().numpy().astype(np.float64)
        for e, d in enumerate(ds):
            d.qpos[:] = q[e].astype(np.float32); d.qvel[:] = v[e].astype(np.float32); d.ctrl[:] = ctrl[e]
            d.qacc_warmstart[:] = 0 if k == 0 else d.qacc_warmstart
            d.step(1)
            worst = max(worst, float(np.abs(vg[e] - d.qvel).max() / max(1.0, np.abs(d.qvel).max())))
            assert d.nefc >= 4
            q[e] = d.qpos; v[e] = d.qvel
    assert worst < 2e-4, worst
    assert int(st.status.max()) == 0
    # activation states of the filter / integrator actuators ran free on both sides for 60 steps
    assert cm.na == 2
    np.testing.assert_allclose(st.act.cpu().numpy(), np.array([d.act for d in ds]), rtol=0, atol=2e-5)
    assert float(np.abs(st.act.cpu().numpy()).max()) > 1e-3


@pytest.mark.gpu
def test_tendon_limit_rows_gpu_match_oracle(oracle_lib):
    """tendon_limit_toy (two length-limited spatial tendons + a limited fixed tendon on the elbow): teacher-forced per-substep
    parity of the fused kernel against the oracle; all three limits become active somewhere in the sweep."""
    import torch
    from myosuite_amd import engine as E
    cm = synth.get_model("tendon_limit_toy")
    assert cm.njmax == 4 and int(cm.arrays["TENDON_LIMITED"].sum()) == 3
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    n = 32
    rng = np.random.default_rng(6)
    q = rng.uniform(0.0, 2.27, (n, 1)); v = rng.standard_normal((n, 1)) * 2.0
    act = rng.random((n, cm.na))
    st = E.BatchState(hm, n)
    ds = [O.OracleData(om) for _ in range(n)]
    dv = E.Derived(hm, n, ["nefc"])
    worst, rows = 0.0, set()
    for k in range(40):
        ctrl = rng.random((n, cm.nu)).astype(np.float32)
        st.qpos.copy_(torch.from_numpy(q.astype(np.float32))); st.qvel.copy_(torch.from_numpy(v.astype(np.float32)))
        st.act.copy_(torch.from_numpy(act.astype(np.float32)))
        c = torch.from_numpy(ctrl).cuda()
        E.forward(hm, st, c, dv)
        E.step(hm, st, c, 1)
        vg = st.qvel.cpu().numpy().astype(np.float64)
        for e, d in enumerate(ds):
            d.qpos[:] = q[e].astype(np.float32); d.qvel[:] = v[e].astype(np.float32); d.act[:] = act[e].astype(np.float32); d.ctrl[:] = ctrl[e]
            d.forward()
            assert int(dv["nefc"][e]) == d.nefc
            rows.add(d.nefc)
            d.step(1)
            worst = max(worst, float(np.abs(vg[e] - d.qvel).max() / max(1.0, np.abs(d.qvel).max())))
            q[e] = d.qpos; v[e] = d.qvel; act[e] = d.act
        if k % 10 == 9:                                  # re-spread the states over the joint range
            q = rng.uniform(0.0, 2.27, (n, 1)); v = rng.standard_normal((n, 1)) * 2.0
    assert worst < 2e-4, worst
    assert max(rows) >= 2 and int(st.status.max()) == 0


def test_self_colliding_hand_model_has_finger_contacts(oracle_lib):
    """synth.make_hand(self_collision=True): capsules along the hand's segments and 20 explicit finger / thumb / palm pairs
    (docs/source/suite.rst:288); co-contraction curls the fingers into each other and the oracle reports contacts."""
    cm = synth.get_model("hand_contact"); base = synth.get_model("hand")
    assert (cm.nq, cm.nv, cm.nu) == (base.nq, base.nv, base.nu) and cm.npair == 20 and cm.njmax <= 64
    np.testing.assert_array_equal(cm.arrays["ACT_GAINPRM"], base.arrays["ACT_GAINPRM"])
    om = O.OracleModel(cm)
    rng = np.random.default_rng(0)
    seen = 0
    for trial in range(4):
        d = O.OracleData(om); d.reset(); d.ctrl[:] = rng.random(cm.nu)
        for _ in range(40):
            d.step(10)
            seen = max(seen, d.ncon)
        assert d.warn == 0 and np.all(np.isfinite(d.qpos))
    assert seen >= 2


@pytest.mark.gpu
def test_self_colliding_hand_gpu_matches_oracle(oracle_lib):
    """The Pose task on the self-colliding hand (registry.make(..., model="hand_contact")): teacher-forced env-steps against
    the oracle env while fingers are in contact, and a finite, overflow-free full-size batch."""
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry
    from oracle import env_oracle as EO
    n, nsteps = 12, 10
    env = registry.make("myoHandPoseRandom-v0", num_envs=n, seed=2, autoreset=False, model="hand_contact")
    cm = env.cm
    assert cm.npair == 20 and env.hm.launch_lanes(n) == 64
    env.reset(seed=2)
    orc = []
    for e in range(n):
        o = EO.PoseEnvOracle(cm, pose_thd=env.pose_thd)
        o.reset(env.state.qpos[e].cpu().numpy(), env.target_jnt_value[e].cpu().numpy())
        orc.append(o)
    a = torch.empty(n, cm.nu, device="cuda")
    ncon = 0
    for s in range(nsteps):
        st = env.get_env_state()
        for e in range(n):                       # teacher-forced per env-step (contact onsets are discontinuous)
            d = orc[e].d
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                v = getattr(d, k).astype(np.float32); getattr(d, k)[:] = v
                st[k][e] = torch.from_numpy(v)
        env.set_env_state(st)
        E.uniform(a, 41, s)
        act = (0.5 + 0.5 * a).contiguous()       # strong co-contraction: the fingers curl into each other
        obs, r, term, trunc, info = env.step(act)
        an = act.cpu().numpy()
        for e in range(n):
            ob, rr, done, rd = orc[e].step(an[e].astype(np.float64))
            ncon = max(ncon, orc[e].d.ncon)
            got = obs[e].cpu().numpy()
            tol = np.full(got.shape, 2e-3); tol[cm.nq:cm.nq + cm.nv] = 1e-2
            bad = np.abs(got - ob) / np.maximum(1.0, np.abs(ob)) > tol
            assert not bad.any(), (s, e, np.nonzero(bad)[0][:5], np.abs(got - ob)[bad][:5])
            assert abs(float(r[e]) - rr) < 5e-3 * max(1.0, abs(rr))
    assert ncon >= 1
    big = registry.make("myoHandPoseRandom-v0", num_envs=4096, seed=3, model="hand_contact")
    big.rollout_setup(action_seed=5)
    for s in range(30):
        obs, rwd, mask = big.rollout_step(None, stream_id=s)
    assert bool(torch.isfinite(obs).all()) and int((big.state.status & 0xA).max()) == 0


def test_hand_dense_pair_list_is_longer_than_a_wave(oracle_lib):
    """synth.make_hand_dense(): the reorient hand under MuJoCo's default collision filter (every skin capsule against every other
    one that is not on the same body or on parent and child, `myohand_sar.xml:15-18`) -- a pair list of ~190 entries, three times
    the 64 lanes of a wave.  Structure of the list, and the oracle sees finger-finger AND object contacts."""
    from myosuite_amd.model.spec import C
    cm = synth.get_model("hand_dense"); base = synth.get_model("hand_reorient")
    assert (cm.nq, cm.nv, cm.nu) == (base.nq, base.nv, base.nu)
    assert 120 <= cm.npair <= C["MM_MAX_PAIRS"] and cm.njmax <= 64, cm.npair
    g1, g2 = cm.arrays["PAIR_GEOM1"], cm.arrays["PAIR_GEOM2"]
    gb, par = cm.arrays["GEOM_BODYID"], cm.arrays["BODY_PARENT"]
    b1, b2 = gb[g1], gb[g2]
    assert np.all(b1 != b2) and np.all(par[b1] != b2) and np.all(par[b2] != b1)            # mj_collision's filter
    key = np.minimum(b1, b2).astype(np.int64) * 1000 + np.maximum(b1, b2)
    assert np.all(np.diff(key) >= 0)                                                          # body-pair order
    obj = cm.names["geom"]["obj"]
    om = O.OracleModel(cm)
    rng = np.random.default_rng(1)
    self_seen = obj_seen = 0
    for trial in range(6):
        d = O.OracleData(om); d.reset(); d.qpos[0] = -1.5; d.ctrl[:] = 0.5 + 0.5 * rng.random(cm.nu)
        for _ in range(30):
            d.step(5)
            for p in d.con_pair[:d.ncon]:
                if obj in (int(g1[p]), int(g2[p])): obj_seen += 1
                else: self_seen += 1
        assert np.all(np.isfinite(d.qpos))
    assert self_seen >= 1 and obj_seen >= 1, (self_seen, obj_seen)


@pytest.mark.gpu
@pytest.mark.parametrize("nconmax", [0, 2], ids=["model-bounds", "nconmax2"])
def test_gpu_pair_list_longer_than_the_wave_matches_oracle(oracle_lib, nconmax):
    """The explicit pair list is swept in chunks of one pair per lane (make_constraint_gen): contacts found in different chunks must
    be numbered, bounded by nconmax and laid out as rows in PAIR order, exactly as the oracle's loop over the list does.  States: a
    random-action rollout of the reorient task on `hand_dense` (189 pairs = three chunks at 64 lanes); every env's row count and
    constrained acceleration against the oracle -- with the model's own bounds, and with nconmax cut to 2 so that the drop falls
    across chunk boundaries in most envs (both sides flag it: status bit 8 / oracle warning)."""
    import torch
    from myosuite_amd import engine as E
    from myosuite_amd.envs import registry
    n = 512
    env = registry.make("myoHandReorient100-v0", num_envs=n, seed=5, model="hand_dense")
    assert env.cm.npair > 128 and env.hm.launch_lanes(n) == 64
    env.rollout_setup(action_seed=11)
    for s in range(25):
        env.rollout_step(None, stream_id=s)
    st0 = env.state
    cm = env.cm if nconmax == 0 else synth.compile_spec("hand_dense", edit=lambda sp: setattr(sp, "nconmax", nconmax))
    hm = env.hm if nconmax == 0 else E.HipModel(cm, lanes_per_env=64)
    st = E.BatchState(hm, n)
    for k in ("qpos", "qvel", "act", "qacc_warmstart"):
        getattr(st, k).copy_(getattr(st0, k))
    st.set_geom_size_env(int(st0._c.geom_env_id), st0.geom_size_env.clone())
    st.set_geom_type_env(st0.geom_type_env.clone())
    if st0.body_mass_env is not None:
        st.set_body_mass_env(int(st0._c.body_mass_env_id), st0.body_mass_env.clone())
    ctrl = env.last_ctrl.clone()
    d_ = E.Derived(hm, n, ["qacc", "nefc"])
    E.forward(hm, st, ctrl, d_)
    torch.cuda.synchronize()
    ga, gn = d_["qacc"].cpu().numpy().astype(np.float64), d_["nefc"].cpu().numpy()
    gstat = st.status.cpu().numpy()
    om = O.OracleModel(cm); d = O.OracleData(om)
    qpos, qvel, act, warm = (getattr(st, k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "act", "qacc_warmstart"))
    gs, gt, c = st.geom_size_env.cpu().numpy().astype(np.float64), st.geom_type_env.cpu().numpy(), ctrl.cpu().numpy().astype(np.float64)
    bm = st.body_mass_env.cpu().numpy().astype(np.float64) if st.body_mass_env is not None else None
    mism = dropped = multi = deep = 0
    rel = []
    chunk_of = lambda p: int(p) // 64
    g1, g2, gsz, obj_g = cm.arrays["PAIR_GEOM1"], cm.arrays["PAIR_GEOM2"], cm.arrays["GEOM_SIZE"].reshape(-1, 3), int(st._c.geom_env_id)
    for e in range(n):
        d.reset()                                                                     # (clears the sticky warning bits)
        d.set_geom_size(int(st._c.geom_env_id), gs[e], int(gt[e]))
        if bm is not None:
            d.set_body_mass(int(st._c.body_mass_env_id), float(bm[e]))
        d.qpos[:] = qpos[e]; d.qvel[:] = qvel[e]; d.act[:] = act[e]; d.ctrl[:] = c[e]; d.qacc_warmstart[:] = warm[e]
        d.forward()
        if len({chunk_of(p) for p in d.con_pair[:d.ncon]}) > 1:
            multi += 1
        odrop = bool(int(d.warn) & 6)
        dropped += odrop
        assert odrop == bool(gstat[e] & 8), (e, int(d.warn), int(gstat[e]))
        if d.nefc != gn[e]:
            mism += 1
            continue
        # (a skin capsule pushed in by more than three quarters of its radius -- its AXIS at or inside the surface of the cylinder / box / ellipsoid object: the one class where a
        #  closest-feature collider is ill-conditioned in fp32 and fp64 alike -- tests/test_fuzz_models.py; counted, not compared)
        if int(gt[e]) >= 4 and any(obj_g in (int(g1[p]), int(g2[p])) and float(d.con_dist[c_]) < -0.75 * float(gsz[int(g1[p]) if int(g2[p]) == obj_g else int(g2[p]), 0])
                                    for c_, p in enumerate(d.con_pair[:d.ncon])) and \
                np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max()) > 1e-3:
            deep += 1                         # (forgiven only when it is actually off, and only a handful: asserted below)
            continue
        rel.append(np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max()))
    rel = np.array(rel)
    print(f"hand_dense nconmax={nconmax or cm.nconmax}: {n} envs, contacts in more than one chunk {multi}, envs with a drop {dropped}, "
          f"row-count mismatches {mism}, deep capsule-in-convex envs that are off (left out) {deep}, rel |dqacc| median {np.median(rel):.1e} max {rel.max():.1e}")
    assert deep <= 2 + n // 250, deep
    assert multi >= n // 20, multi
    if nconmax:
        assert dropped >= n // 50, dropped
    assert mism <= max(2, n // 100) and rel.max() < 2e-3 and np.quantile(rel, 0.99) < 3e-4, (mism, rel.max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,family", [("star", 1), ("chain", 2), ("comb", 1), ("free", 1)])
def test_dof_tree_shapes_route_and_match_oracle(oracle_lib, kind, family):
    """tree_toy models (limit rows only; dof trees of different shapes): the star and the comb (branches of unequal length:
    more elimination steps) take the tree-sparse L'DL kernels (one lane per segment of the dof tree), the 10-deep chain the
    general-row kernels with the dense Cholesky -- and all three agree with the oracle, stage by stage (M, qacc_smooth, qacc) and over a free run with joint
    limits coming and going."""
    import torch
    from myosuite_amd import engine as E
    cm = synth.get_model(f"tree_{kind}")
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    assert hm.info(E.INFO_KERNEL_FAMILY) == family
    n = 24
    rng = np.random.default_rng(8)
    q = rng.uniform(-0.9, 0.8, (n, cm.nq)); v = rng.standard_normal((n, cm.nv)) * 1.5
    if kind == "free":
        q[:, :3] = [0.0, 0.0, 1.0]; q[:, 3:7] /= np.linalg.norm(q[:, 3:7], axis=1, keepdims=True)
    st = E.BatchState(hm, n)
    ds = [O.OracleData(om) for _ in range(n)]
    worst, rows = 0.0, set()
    for k in range(50):
        ctrl = rng.uniform(-1, 1, (n, cm.nu)).astype(np.float32)
        st.qpos.copy_(torch.from_numpy(q.astype(np.float32))); st.qvel.copy_(torch.from_numpy(v.astype(np.float32)))
        c = torch.from_numpy(ctrl).cuda()
        if k % 10 == 0:
            dump = E.debug_dump(hm, st, c).cpu().numpy().astype(np.float64)
        E.step(hm, st, c, 1)
        vg = st.qvel.cpu().numpy().astype(np.float64)
        for e, d in enumerate(ds):
            d.qpos[:] = q[e].astype(np.float32); d.qvel[:] = v[e].astype(np.float32); d.ctrl[:] = ctrl[e]
            if k % 10 == 0:
                d.forward()
                M = dump[e, hm.layout("M"):hm.layout("M") + cm.nv * cm.nv].reshape(cm.nv, cm.nv)
                assert np.abs(M - d.full_M()).max() < 2e-5 * np.abs(d.full_M()).max()
                for nm, ref in (("qaccsm", d.qacc_smooth), ("qacc", d.qacc)):
                    got = dump[e, hm.layout(nm):hm.layout(nm) + cm.nv]
                    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), (nm, e)
            d.step(1)
            rows.add(d.nefc)
            worst = max(worst, float(np.abs(vg[e] - d.qvel).max() / max(1.0, np.abs(d.qvel).max())))
            q[e] = d.qpos; v[e] = d.qvel
    assert worst < 2e-4, worst
    assert len(rows) >= 3 and int(st.status.max()) == 0
