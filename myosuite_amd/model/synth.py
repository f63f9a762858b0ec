"""Synthetic-but-structurally-faithful musculoskeletal models.

The reference's real MJCF (``myosuite/simhive/myo_sim``) is an empty submodule in
/root/reference, so the task XMLs that only ``<include>`` it
(myosuite/envs/myo/assets/elbow/myoelbow_1dof6muscles.xml:10-13,
.../hand/myohand_pose.xml:10-15) cannot be compiled.  These generators build
models with the DIMENSIONS, joint names/order and muscle names that in-repo
evidence pins (joint names myosuite/envs/myo/myobase/__init__.py:301-325;
muscle names docs/source/suite.rst:75-80,99-119) and the same feature mix:
hinge trees, spatial tendons with via-points and sphere/cylinder wrapping with
side-sites, Hill-type muscles (MuJoCo muscle defaults), limited joints with
damping/armature, timestep 0.002.
"""
from __future__ import annotations

import math

import numpy as np

from .spec import ModelSpec, CompiledModel

QX90N = (math.cos(-math.pi / 4), math.sin(-math.pi / 4), 0.0, 0.0)  # local z -> world y


def _cyl_inertia(m, r, L, axis="x"):
    ia = 0.5 * m * r * r
    it = m * (3 * r * r + L * L) / 12.0
    return {"x": (ia, it, it), "y": (it, ia, it), "z": (it, it, ia)}[axis]


# ----------------------------------------------------------------------------- elbow
def make_elbow() -> ModelSpec:
    """myoElbow: 1 DoF (r_elbow_flex), 6 muscles (TRIlong TRIlat TRImed BIClong BICshort BRA)."""
    s = ModelSpec("myoelbow_1dof6muscles")
    s.add_body("humerus", "world", pos=(0, 0, 1.2), mass=2.0, ipos=(0, 0, -0.15),
               inertia=_cyl_inertia(2.0, 0.03, 0.30, "z"))
    s.add_body("forearm", "humerus", pos=(0, 0, -0.30), mass=1.5, ipos=(0, 0, -0.14),
               inertia=_cyl_inertia(1.5, 0.03, 0.30, "z"))
    s.add_joint("r_elbow_flex", "forearm", "hinge", pos=(0, 0, 0), axis=(0, -1, 0), range=(0.0, 2.27),
                damping=0.4, armature=0.01, stiffness=0.0)
    # wrap cylinder on the elbow axis (humerus frame), axis along world y
    s.add_geom("elbow_wrap", "humerus", "cylinder", size=(0.022, 0.05), pos=(0, 0, -0.30), quat=QX90N)
    s.add_site("elbow_side_post", "humerus", (-0.06, 0, -0.30))
    s.add_site("elbow_side_ant", "humerus", (0.06, 0, -0.30))
    s.add_site("wrist", "forearm", (0, 0, -0.27))
    s.add_site("wrist_target", "world", (0.001, 0.001, 0.001))

    def tri(name, y, org, force):
        s.add_site(f"{name}_o", "humerus", org)
        s.add_site(f"{name}_v", "humerus", (-0.028, y, -0.24))
        s.add_site(f"{name}_i", "forearm", (-0.022, y, 0.018))
        s.add_tendon(f"{name}_tendon", [("site", f"{name}_o"), ("site", f"{name}_v"),
                                        ("cylinder", "elbow_wrap", "elbow_side_post"), ("site", f"{name}_i")])
        s.add_muscle(name, f"{name}_tendon", force=force)

    def flex(name, y, org, via, ins, force, wrap=True):
        s.add_site(f"{name}_o", "humerus", org)
        s.add_site(f"{name}_v", "humerus", via)
        s.add_site(f"{name}_i", "forearm", ins)
        path = [("site", f"{name}_o"), ("site", f"{name}_v")]
        if wrap:
            path.append(("cylinder", "elbow_wrap", "elbow_side_ant"))
        path.append(("site", f"{name}_i"))
        s.add_tendon(f"{name}_tendon", path)
        s.add_muscle(name, f"{name}_tendon", force=force)

    tri("TRIlong", -0.006, (-0.025, -0.006, -0.02), 800.0)
    tri("TRIlat", 0.008, (-0.02, 0.008, -0.09), 620.0)
    tri("TRImed", 0.0, (-0.015, 0.0, -0.14), 620.0)
    flex("BIClong", -0.004, (0.02, -0.004, -0.01), (0.032, -0.004, -0.22), (0.014, -0.004, -0.05), 620.0)
    flex("BICshort", 0.006, (0.025, 0.006, -0.03), (0.03, 0.006, -0.21), (0.014, 0.006, -0.055), 430.0)
    flex("BRA", 0.0, (0.012, 0.0, -0.18), (0.02, 0.0, -0.25), (0.022, 0.0, -0.20), 260.0, wrap=False)
    return s


# ----------------------------------------------------------------------------- elbow + exo
def make_elbow_exo() -> ModelSpec:
    """myoElbow with the 1-DoF exoskeleton of assets/elbow/myoelbow_1dof6muscles_1dofexo.xml: the same arm plus a torque
    actuator on the elbow joint (after the six muscles: nu = 7, na = 6) and the ``carry_weight`` body whose mass the Random
    variant re-draws per episode (pose_v0.py:177-187)."""
    s = make_elbow()
    s.name = "myoelbow_1dof6muscles_1dofexo"
    s.add_body("carry_weight", "forearm", pos=(0, 0, -0.27), mass=1.0, inertia=(4e-4, 4e-4, 4e-4))
    s.add_motor("Exo", "r_elbow_flex", gear=10.0, ctrlrange=(-1.0, 1.0))
    return s


# ----------------------------------------------------------------------------- finger
FINGER_JOINTS = ["IFadb", "IFmcp", "IFpip", "IFdip"]
FINGER_MUSCLES = ["EXTN", "adabR", "adabL", "mflx", "dflx"]          # docs/source/suite.rst:48-56


def make_finger(motor: bool = False) -> ModelSpec:
    """myoFinger / motorFinger (simhive/myo_sim/finger, absent from the reference checkout): 4 DoF (IFadb IFmcp IFpip IFdip,
    names from myobase/__init__.py:196-201) and either 5 simplified antagonistic muscle-tendon units (EXTN adabR adabL mflx
    dflx) or 4 joint torque motors ("its robotic counterpart with simple torque actuators", suite.rst:39).  +x distal,
    +z dorsal; positive flexion curls toward -z.  Sites ``IFtip`` / ``IFtip_target`` for the reach task."""
    s = ModelSpec("motorfinger_v0" if motor else "myofinger_v0")
    L = (0.070, 0.045, 0.035)
    s.add_body("metacarpal", "world", pos=(0.0, 0.0, 0.20), mass=0.2, ipos=(0.04, 0, 0),
               inertia=_cyl_inertia(0.2, 0.012, 0.08, "x"))
    s.add_body("mcp_link", "metacarpal", pos=(0.08, 0, 0), mass=0.004, inertia=(4e-7, 4e-7, 4e-7))
    s.add_joint("IFadb", "mcp_link", "hinge", axis=(0, 0, 1), range=(-0.262, 0.262), damping=0.02, armature=0.0008)
    s.add_body("proximal", "mcp_link", pos=(0, 0, 0), mass=0.030, ipos=(0.5 * L[0], 0, 0),
               inertia=_cyl_inertia(0.030, 0.009, L[0], "x"))
    s.add_joint("IFmcp", "proximal", "hinge", axis=(0, 1, 0), range=(-0.785, 1.571), damping=0.02, armature=0.0008)
    s.add_body("middle", "proximal", pos=(L[0], 0, 0), mass=0.015, ipos=(0.5 * L[1], 0, 0),
               inertia=_cyl_inertia(0.015, 0.008, L[1], "x"))
    s.add_joint("IFpip", "middle", "hinge", axis=(0, 1, 0), range=(0.0, 1.571), damping=0.015, armature=0.0006)
    s.add_body("distal", "middle", pos=(L[1], 0, 0), mass=0.008, ipos=(0.5 * L[2], 0, 0),
               inertia=_cyl_inertia(0.008, 0.007, L[2], "x"))
    s.add_joint("IFdip", "distal", "hinge", axis=(0, 1, 0), range=(0.0, 1.571), damping=0.01, armature=0.0005)
    s.add_site("IFtip", "distal", (L[2], 0, 0))
    s.add_site("IFtip_target", "world", (0.2, 0.05, 0.20))
    if motor:
        for j, gear in zip(FINGER_JOINTS, (0.15, 0.4, 0.25, 0.15)):
            s.add_motor("A_" + j, j, gear=gear, ctrlrange=(-1.0, 1.0))
        return s
    # wrap cylinders on the three flexion axes (axis along y), dorsal / palmar side-sites
    for nm, body, r in (("mcp", "metacarpal", 0.010), ("pip", "proximal", 0.008), ("dip", "middle", 0.0065)):
        x = {"mcp": 0.08, "pip": L[0], "dip": L[1]}[nm]
        s.add_geom(f"{nm}_wrap", body, "cylinder", size=(r, 0.02), pos=(x, 0, 0), quat=QX90N)
        s.add_site(f"{nm}_dors", body, (x, 0, 0.03))
        s.add_site(f"{nm}_palm", body, (x, 0, -0.03))
    # EXTN: dorsal, spans mcp / pip / dip
    for nm, body, pos in (("EXTN_o", "metacarpal", (0.02, 0, 0.012)), ("EXTN_a", "metacarpal", (0.06, 0, 0.013)),
                          ("EXTN_b", "proximal", (0.035, 0, 0.011)), ("EXTN_c", "middle", (0.022, 0, 0.009)),
                          ("EXTN_i", "distal", (0.012, 0, 0.0075))):
        s.add_site(nm, body, pos)
    s.add_tendon("EXTN_tendon", [("site", "EXTN_o"), ("site", "EXTN_a"), ("cylinder", "mcp_wrap", "mcp_dors"),
                                 ("site", "EXTN_b"), ("cylinder", "pip_wrap", "pip_dors"), ("site", "EXTN_c"),
                                 ("cylinder", "dip_wrap", "dip_dors"), ("site", "EXTN_i")])
    s.add_muscle("EXTN", "EXTN_tendon", force=120.0)
    # abduction pair: lateral bands crossing the mcp on either side
    for nm, y in (("adabR", -0.012), ("adabL", 0.012)):
        s.add_site(f"{nm}_o", "metacarpal", (0.045, 1.4 * y, -0.002))
        s.add_site(f"{nm}_i", "proximal", (0.018, y, -0.003))
        s.add_tendon(f"{nm}_tendon", [("site", f"{nm}_o"), ("site", f"{nm}_i")])
        s.add_muscle(nm, f"{nm}_tendon", force=60.0)
    # mflx: palmar, inserts on the middle phalanx (mcp + pip); dflx: palmar, inserts on the distal phalanx
    for nm, body, pos in (("mflx_o", "metacarpal", (0.02, -0.003, -0.012)), ("mflx_a", "metacarpal", (0.06, -0.003, -0.0135)),
                          ("mflx_b", "proximal", (0.035, -0.003, -0.011)), ("mflx_i", "middle", (0.015, -0.003, -0.009)),
                          ("dflx_o", "metacarpal", (0.02, 0.003, -0.014)), ("dflx_a", "metacarpal", (0.06, 0.003, -0.0145)),
                          ("dflx_b", "proximal", (0.035, 0.003, -0.012)), ("dflx_c", "middle", (0.022, 0.003, -0.0095)),
                          ("dflx_i", "distal", (0.012, 0.003, -0.0075))):
        s.add_site(nm, body, pos)
    s.add_tendon("mflx_tendon", [("site", "mflx_o"), ("site", "mflx_a"), ("cylinder", "mcp_wrap", "mcp_palm"),
                                 ("site", "mflx_b"), ("cylinder", "pip_wrap", "pip_palm"), ("site", "mflx_i")])
    s.add_muscle("mflx", "mflx_tendon", force=110.0)
    s.add_tendon("dflx_tendon", [("site", "dflx_o"), ("site", "dflx_a"), ("cylinder", "mcp_wrap", "mcp_palm"),
                                 ("site", "dflx_b"), ("cylinder", "pip_wrap", "pip_palm"), ("site", "dflx_c"),
                                 ("cylinder", "dip_wrap", "dip_palm"), ("site", "dflx_i")])
    s.add_muscle("dflx", "dflx_tendon", force=110.0)
    return s


# ----------------------------------------------------------------------------- torso
TORSO_JOINTS = ["flex_extension", "lat_bending", "axial_rotation", "Abs_t1", "Abs_t2", "Abs_r3",
                "L4_L5_FE", "L4_L5_LB", "L4_L5_AR", "L3_L4_FE", "L3_L4_LB", "L3_L4_AR",
                "L2_L3_FE", "L2_L3_LB", "L2_L3_AR", "L1_L2_FE", "L1_L2_LB", "L1_L2_AR"]      # myobase/__init__.py:645-664
TORSO_GROUPS = [("rect_abd", 1), ("IL", 12), ("QL", 18), ("MF", 25), ("LT", 19), ("EO", 6), ("IO", 6), ("PS", 11),
                ("LD", 7)]                                                                    # fascicles per side: 105


def make_torso(exosuit: bool = False) -> ModelSpec:
    """myoTorso: "210 actuators and 18 joints" (docs/source/suite.rst:207), generated from OpenSim's *constrained* lumbar spine
    model: the pelvis is fixed, the three L5/S1 rotations (flex_extension, lat_bending, axial_rotation) drive the L4-L5 ...
    L1-L2 rotations and the abdomen's Abs_t1/Abs_t2/Abs_r3 through joint equalities (here: linear couplings).  210 muscle
    fascicles = 105 per side in the groups of suite.rst:213-224 (rect_abd, IL, QL, MF, LT, EO, IO + psoas / latissimus);
    straight or one-via-point paths from pelvis / sacrum to the vertebrae and the rib cage.  +x anterior, +y left, +z up."""
    s = ModelSpec("myotorso")
    s.timestep = 0.002
    s.add_body("pelvis", "world", pos=(0, 0, 0.95), mass=10.0, inertia=(0.09, 0.08, 0.09))
    H = 0.036                                     # vertebral spacing
    lv = ["lumbar5", "lumbar4", "lumbar3", "lumbar2", "lumbar1"]
    jn = [("flex_extension", "lat_bending", "axial_rotation")] + [(f"L{5 - k}_L{6 - k}_FE", f"L{5 - k}_L{6 - k}_LB", f"L{5 - k}_L{6 - k}_AR")
                                                                   for k in range(1, 5)]
    # joint order must follow TORSO_JOINTS: lumbar5 (3), abdomen (3), then lumbar4..lumbar1
    def vertebra(k, parent):
        top = k == 4
        m = 14.0 if top else 1.8                  # lumbar1 carries the rib cage / thorax
        s.add_body(lv[k], parent, pos=(0, 0, 0.05 if k == 0 else H), mass=m, ipos=(0.01, 0, 0.20 if top else 0.0),
                   inertia=(0.55, 0.45, 0.25) if top else (0.006, 0.006, 0.009))
        lim = (0.5, 0.35, 0.3) if k == 0 else (0.2, 0.12, 0.1)
        for name, ax, r in zip(jn[k], ((0, 1, 0), (1, 0, 0), (0, 0, 1)), lim):
            s.add_joint(name, lv[k], "hinge", axis=ax, range=(-r, r), damping=5.0 if k == 0 else 2.0, armature=0.05 if k == 0 else 0.25,
                        stiffness=150.0 if k == 0 else 25.0)
    vertebra(0, "pelvis")
    s.add_body("abdomen", "pelvis", pos=(0.07, 0, 0.12), mass=6.0, inertia=(0.06, 0.05, 0.06))
    s.add_joint("Abs_t1", "abdomen", "slide", axis=(1, 0, 0), range=(-0.08, 0.08), damping=30.0, armature=0.5)
    s.add_joint("Abs_t2", "abdomen", "slide", axis=(0, 0, 1), range=(-0.08, 0.08), damping=30.0, armature=0.5)
    s.add_joint("Abs_r3", "abdomen", "hinge", axis=(0, 1, 0), range=(-0.8, 0.8), damping=1.0, armature=0.05)
    for k in range(1, 5):
        vertebra(k, lv[k - 1])
    assert [j.name for j in s.joints] == TORSO_JOINTS
    # constrained spine: fixed fractions of the net rotation at each level, abdomen follows the flexion
    for k, fr in zip(range(1, 5), (0.26, 0.19, 0.14, 0.10)):
        for c in range(3):
            s.add_equality_joint(jn[k][c], jn[0][c], (0.0, fr))
    s.add_equality_joint("Abs_t1", "flex_extension", (0.0, 0.035))
    s.add_equality_joint("Abs_t2", "flex_extension", (0.0, -0.012))
    s.add_equality_joint("Abs_r3", "flex_extension", (0.0, 0.45))
    # ---- 210 fascicles
    side_sign = {"r": -1.0, "l": 1.0}
    attach = {  # group: (origin body, origin xyz (right side, y<0 mirrored), insertion bodies cycled, insertion xyz, via?, Fmax)
        "rect_abd": ("pelvis", (0.085, 0.035, -0.03), ["lumbar1"], (0.125, 0.04, 0.24), ("abdomen", (0.055, 0.04, 0.0)), 410.0),
        "IL": ("pelvis", (-0.075, 0.07, 0.03), ["lumbar1"], (-0.055, 0.085, 0.16), None, 160.0),
        "QL": ("pelvis", (-0.045, 0.085, 0.035), ["lumbar4", "lumbar3", "lumbar2", "lumbar1"], (-0.025, 0.04, 0.0), None, 60.0),
        "MF": ("pelvis", (-0.07, 0.02, 0.01), lv, (-0.045, 0.012, 0.008), None, 70.0),
        "LT": ("pelvis", (-0.08, 0.045, 0.02), ["lumbar1", "lumbar2", "lumbar3"], (-0.06, 0.05, 0.10), None, 120.0),
        "EO": ("pelvis", (0.06, 0.115, 0.01), ["lumbar1"], (0.07, 0.12, 0.14), ("abdomen", (0.03, 0.13, 0.0)), 200.0),
        "IO": ("pelvis", (0.02, 0.12, 0.02), ["lumbar1"], (0.09, 0.07, 0.12), ("abdomen", (0.04, 0.11, 0.01)), 180.0),
        "PS": ("pelvis", (0.035, 0.075, -0.07), lv, (0.018, 0.028, 0.0), None, 150.0),
        "LD": ("pelvis", (-0.07, 0.06, 0.03), ["lumbar1"], (-0.02, 0.15, 0.30), None, 110.0),
    }
    rng = np.random.default_rng(7)
    jit_store = {(grp, f): rng.uniform(-1, 1, 6) * 0.006 for grp, cnt in TORSO_GROUPS for f in range(cnt)}   # mirrored l/r
    nm = 0
    for side in ("r", "l"):
        sg = side_sign[side]
        for grp, cnt in TORSO_GROUPS:
            ob, op, ins_bodies, ip, via, fmax = attach[grp]
            for f in range(cnt):
                name = f"{grp}{f + 1}_{side}" if cnt > 1 else f"{grp}_{side}"
                jit = jit_store[(grp, f)]
                ib = ins_bodies[f % len(ins_bodies)]
                spread = (f / max(cnt - 1, 1) - 0.5)
                o = (op[0] + jit[0], sg * (op[1] + 0.012 * spread + abs(jit[1])), op[2] + jit[2] + 0.01 * spread)
                i = (ip[0] + jit[3], sg * (ip[1] + 0.008 * spread + abs(jit[4])), ip[2] + jit[5])
                s.add_site(name + "_o", ob, o)
                path = [("site", name + "_o")]
                if via is not None:
                    vb, vp = via
                    s.add_site(name + "_v", vb, (vp[0] + jit[3], sg * (vp[1] + abs(jit[4])), vp[2] + jit[5]))
                    path.append(("site", name + "_v"))
                s.add_site(name + "_i", ib, i)
                path.append(("site", name + "_i"))
                s.add_tendon(name + "_tendon", path)
                s.add_muscle(name, name + "_tendon", force=fmax * (0.7 + 0.6 * (f % 3) / 2.0))
                nm += 1
    assert nm == 210
    if exosuit:
        # back exosuit (simhive/myo_sim/torso/myotorso_exosuit.xml, absent from the reference checkout): two elastic cables
        # from the back of the pelvis to the thorax that stretch in flexion (spring beyond their slack length, damped), each
        # with a cable-tension actuator (a tendon-transmission <general>, no activation state) behind the 210 muscles
        for side, sg in (("r", -1.0), ("l", 1.0)):
            s.add_site(f"exo_o_{side}", "pelvis", (-0.11, sg * 0.06, 0.02))
            s.add_site(f"exo_v_{side}", "lumbar3", (-0.075, sg * 0.055, 0.0))
            s.add_site(f"exo_i_{side}", "lumbar1", (-0.09, sg * 0.06, 0.26))
            s.add_tendon(f"exo_cable_{side}", [("site", f"exo_o_{side}"), ("site", f"exo_v_{side}"), ("site", f"exo_i_{side}")],
                         stiffness=4000.0, damping=20.0, springlength=(0.0, 0.355))
            s.add_general(f"Exo_{side}", tendon=f"exo_cable_{side}", gainprm=(-150.0,), ctrlrange=(0.0, 1.0))
        s.name = "myotorso_exosuit"
    return s


# ----------------------------------------------------------------------------- hand
HAND_JOINTS = ["pro_sup", "deviation", "flexion", "cmc_abduction", "cmc_flexion", "mp_flexion",
               "ip_flexion", "mcp2_flexion", "mcp2_abduction", "pm2_flexion", "md2_flexion",
               "mcp3_flexion", "mcp3_abduction", "pm3_flexion", "md3_flexion", "mcp4_flexion",
               "mcp4_abduction", "pm4_flexion", "md4_flexion", "mcp5_flexion", "mcp5_abduction",
               "pm5_flexion", "md5_flexion"]

HAND_MUSCLES = ["ECRL", "ECRB", "ECU", "FCR", "FCU", "PL", "PT", "PQ", "EIP", "EPL", "EPB", "FPL", "APL",
                "OP", "FDS2", "FDS3", "FDS4", "FDS5", "FDP2", "FDP3", "FDP4", "FDP5", "EDC2", "EDC3",
                "EDC4", "EDC5", "EDM", "RI2", "RI3", "RI4", "RI5", "LU_RB2", "LU_RB3", "LU_RB4", "LU_RB5",
                "UI_UB2", "UI_UB3", "UI_UB4", "UI_UB5"]


def make_hand(self_collision: bool = False) -> ModelSpec:
    """myoHand: 23 DoF, 39 muscle-tendon units, 29 bones.  Frame: +x distal, +z dorsal, -y radial.

    ``self_collision``: the real myoHand collides with itself even in the Pose task ("avoiding self collisions poses
    additional challenges", docs/source/suite.rst:288; the MJX README blames HandReach's scaling on "greater contact
    complexity", envs/myo/mjx/README.md:55).  The flag adds collision capsules along metacarpals / phalanges / carpal row and
    20 explicit capsule-capsule pairs (neighbouring fingers segment by segment, thumb against index / middle, finger tips
    against their own metacarpal), condim 3, at most 10 simultaneous contacts (njmax 63)."""
    s = ModelSpec("myohand_contact" if self_collision else "myohand")
    WX = 0.25  # wrist centre
    s.add_body("ulna", "world", pos=(0, 0, 1.0), mass=0.6, ipos=(0.12, 0.01, 0),
               inertia=_cyl_inertia(0.6, 0.015, 0.25, "x"))
    s.add_body("radius", "ulna", pos=(0, 0, 0), mass=0.5, ipos=(0.13, -0.01, 0),
               inertia=_cyl_inertia(0.5, 0.015, 0.25, "x"))
    s.add_joint("pro_sup", "radius", "hinge", pos=(0, 0, 0), axis=(1, 0, 0), range=(-1.57, 1.57),
                damping=0.15, armature=0.002)
    s.add_body("lunate", "radius", pos=(WX, 0, 0), mass=0.03, ipos=(0.004, 0, 0), inertia=(3e-6, 3e-6, 3e-6))
    s.add_joint("deviation", "lunate", "hinge", axis=(0, 0, 1), range=(-0.175, 0.436), damping=0.12,
                armature=0.002)
    s.add_body("scaphoid", "lunate", pos=(0.002, -0.012, 0), mass=0.01, inertia=(1e-6, 1e-6, 1e-6))
    s.add_body("triquetrum", "lunate", pos=(0.002, 0.014, 0), mass=0.01, inertia=(1e-6, 1e-6, 1e-6))
    s.add_body("capitate", "lunate", pos=(0.012, 0, 0), mass=0.05, ipos=(0.01, 0, 0), inertia=(6e-6, 6e-6, 6e-6))
    s.add_joint("flexion", "capitate", "hinge", axis=(0, 1, 0), range=(-1.22, 1.22), damping=0.12,
                armature=0.002)
    s.add_body("trapezium", "capitate", pos=(0.004, -0.022, -0.004), mass=0.01, inertia=(1e-6, 1e-6, 1e-6))
    s.add_body("trapezoid", "capitate", pos=(0.006, -0.012, 0.002), mass=0.01, inertia=(1e-6, 1e-6, 1e-6))
    s.add_body("hamate", "capitate", pos=(0.006, 0.014, 0), mass=0.01, inertia=(1e-6, 1e-6, 1e-6))

    fj = dict(damping=0.05, armature=0.0015)
    # ---- thumb: axes chosen so negative mp/ip flexion curls toward the palm
    tdir = np.array([0.55, -0.70, -0.46]); tdir /= np.linalg.norm(tdir)
    tflex = np.cross(tdir, np.array([0.0, 0.0, 1.0])); tflex /= np.linalg.norm(tflex)  # flexion axis
    tabd = np.cross(tflex, tdir)
    TL = (0.044, 0.032, 0.026)
    s.add_body("firstmc1", "capitate", pos=(0.006, -0.026, -0.008), mass=0.004, inertia=(5e-7, 5e-7, 5e-7))
    s.add_joint("cmc_abduction", "firstmc1", "hinge", axis=tuple(tabd), range=(-0.209, 0.698), **fj)
    s.add_body("firstmc", "firstmc1", pos=(0, 0, 0), mass=0.012, ipos=tuple(0.5 * TL[0] * tdir),
               inertia=(1.2e-6, 1.2e-6, 1.2e-6))
    s.add_joint("cmc_flexion", "firstmc", "hinge", axis=tuple(-tflex), range=(-0.35, 0.70), **fj)
    s.add_body("proximal_thumb", "firstmc", pos=tuple(TL[0] * tdir), mass=0.008,
               ipos=tuple(0.5 * TL[1] * tdir), inertia=(6e-7, 6e-7, 6e-7))
    s.add_joint("mp_flexion", "proximal_thumb", "hinge", axis=tuple(-tflex), range=(-0.785, 0.262), **fj)
    s.add_body("distal_thumb", "proximal_thumb", pos=tuple(TL[1] * tdir), mass=0.005,
               ipos=tuple(0.5 * TL[2] * tdir), inertia=(3e-7, 3e-7, 3e-7))
    s.add_joint("ip_flexion", "distal_thumb", "hinge", axis=tuple(-tflex), range=(-1.31, 0.262), **fj)
    s.add_site("THtip", "distal_thumb", tuple(TL[2] * tdir))

    # ---- fingers 2..5
    fy = {2: -0.026, 3: -0.006, 4: 0.012, 5: 0.029}
    fmc = {2: 0.066, 3: 0.064, 4: 0.058, 5: 0.053}
    flen = {2: (0.043, 0.025, 0.019), 3: (0.047, 0.029, 0.020), 4: (0.043, 0.027, 0.020), 5: (0.034, 0.020, 0.018)}
    mcname = {2: "secondmc", 3: "thirdmc", 4: "fourthmc", 5: "fifthmc"}
    tipname = {2: "IFtip", 3: "MFtip", 4: "RFtip", 5: "LFtip"}
    MC0 = 0.018
    for k in (2, 3, 4, 5):
        L = flen[k]
        s.add_body(mcname[k], "capitate", pos=(MC0, fy[k], 0), mass=0.02, ipos=(0.5 * fmc[k], 0, 0),
                   inertia=_cyl_inertia(0.02, 0.006, fmc[k], "x"))
        s.add_body(f"proxph{k}", mcname[k], pos=(fmc[k], 0, 0), mass=0.011, ipos=(0.5 * L[0], 0, 0),
                   inertia=_cyl_inertia(0.011, 0.008, L[0], "x"))
        s.add_joint(f"mcp{k}_flexion", f"proxph{k}", "hinge", axis=(0, 1, 0), range=(-0.785, 1.571), **fj)
        s.add_joint(f"mcp{k}_abduction", f"proxph{k}", "hinge", axis=(0, 0, 1), range=(-0.262, 0.262), **fj)
        s.add_body(f"midph{k}", f"proxph{k}", pos=(L[0], 0, 0), mass=0.006, ipos=(0.5 * L[1], 0, 0),
                   inertia=_cyl_inertia(0.006, 0.007, L[1], "x"))
        s.add_joint(f"pm{k}_flexion", f"midph{k}", "hinge", axis=(0, 1, 0), range=(0.0, 1.571), **fj)
        s.add_body(f"distph{k}", f"midph{k}", pos=(L[1], 0, 0), mass=0.004, ipos=(0.5 * L[2], 0, 0),
                   inertia=_cyl_inertia(0.004, 0.006, L[2], "x"))
        s.add_joint(f"md{k}_flexion", f"distph{k}", "hinge", axis=(0, 1, 0), range=(0.0, 1.571), **fj)
        s.add_site(tipname[k], f"distph{k}", (L[2], 0, 0))
        # extensor wrap cylinders at MCP and PIP (axis = flexion axis y)
        s.add_geom(f"mcp{k}_wrap", mcname[k], "cylinder", size=(0.0065, 0.01), pos=(fmc[k], 0, 0), quat=QX90N)
        s.add_site(f"mcp{k}_side", mcname[k], (fmc[k], 0, 0.03))
        s.add_geom(f"pip{k}_wrap", f"proxph{k}", "cylinder", size=(0.0045, 0.008), pos=(L[0], 0, 0), quat=QX90N)
        s.add_site(f"pip{k}_side", f"proxph{k}", (L[0], 0, 0.03))
    for n in ("THtip", "IFtip", "MFtip", "RFtip", "LFtip"):
        s.add_site(n + "_target", "world", (0, 0, 0.002))

    # wrist wrap obstacle: cylinder ON the flexion axis (capitate origin), carried by the lunate.  Via points sit
    # outside the cylinder at every wrist angle and the wrap angle stays < ~80 deg over the (soft) joint range, so
    # tendon lengths are continuous inside the reachable box (checked by tools/check_tendon_continuity.py).
    s.add_geom("wrist_wrap", "lunate", "cylinder", size=(0.013, 0.03), pos=(0.012, 0, 0), quat=QX90N)
    s.add_site("wrist_side_dors", "lunate", (0.012, 0, 0.06))
    s.add_site("wrist_side_palm", "lunate", (0.012, 0, -0.06))
    # thumb MP wrap sphere for the long extensor
    s.add_geom("thmp_wrap", "firstmc", "sphere", size=(0.006,), pos=tuple(TL[0] * tdir))
    s.add_site("thmp_side", "firstmc", tuple(TL[0] * tdir + 0.03 * np.cross(-tflex, tdir)))

    cnt = [0]

    def site(body, p):
        cnt[0] += 1
        n = f"p{cnt[0]}"
        s.add_site(n, body, tuple(float(x) for x in p))
        return ("site", n)

    def finger_path(k, side, last):
        """via points along finger k; side=-1 palmar (flexor) / +1 dorsal (extensor);
        last: 1 -> insert on proximal, 2 -> middle, 3 -> distal phalanx."""
        L = flen[k]
        hm = {-1: (0.010, 0.0075, 0.005), 1: (0.0085, 0.0065, 0.0045)}[side]
        z = lambda i: side * hm[i]
        p = [site(mcname[k], (fmc[k] - 0.012, 0, z(0)))]
        if side > 0:
            p.append(("cylinder", f"mcp{k}_wrap", f"mcp{k}_side"))
        p.append(site(f"proxph{k}", (0.010, 0, z(0))))
        if last == 1:
            return p
        p.append(site(f"proxph{k}", (L[0] - 0.008, 0, z(1))))
        if side > 0:
            p.append(("cylinder", f"pip{k}_wrap", f"pip{k}_side"))
        p.append(site(f"midph{k}", (0.007, 0, z(1))))
        if last == 2:
            return p
        p.append(site(f"midph{k}", (L[1] - 0.006, 0, z(2))))
        p.append(site(f"distph{k}", (0.006, 0, z(2))))
        return p

    def wrist_path(side, y, radial_body="radius"):
        """forearm via, wrist obstacle, carpal via; side -1 palmar / +1 dorsal"""
        p = [site(radial_body, (WX + 0.012 - 0.040, y, side * 0.014))]
        p.append(("cylinder", "wrist_wrap", "wrist_side_dors" if side > 0 else "wrist_side_palm"))
        p.append(site("capitate", (0.022, y, side * 0.014)))
        return p

    def muscle(name, path, force):
        s.add_tendon(name + "_tendon", path)
        # operating range wide enough that stretched antagonists produce passive force
        s.add_muscle(name, name + "_tendon", force=force, range=(0.70, 1.25))

    # --- wrist movers (insert on metacarpal bases) and forearm rotators
    muscle("ECRL", [site("ulna", (0.02, -0.02, 0.01))] + wrist_path(+1, -0.022) + [site("secondmc", (0.004, 0, 0.008))], 300.0)
    muscle("ECRB", [site("ulna", (0.03, -0.015, 0.012))] + wrist_path(+1, -0.008) + [site("thirdmc", (0.004, 0, 0.008))], 250.0)
    muscle("ECU", [site("ulna", (0.04, 0.02, 0.01))] + wrist_path(+1, 0.026) + [site("fifthmc", (0.004, 0, 0.007))], 200.0)
    muscle("FCR", [site("ulna", (0.02, -0.005, -0.015))] + wrist_path(-1, -0.016) + [site("secondmc", (0.004, 0, -0.008))], 250.0)
    muscle("FCU", [site("ulna", (0.03, 0.02, -0.012))] + wrist_path(-1, 0.024) + [site("fifthmc", (0.004, 0, -0.008))], 300.0)
    muscle("PL", [site("ulna", (0.02, 0.004, -0.016))] + wrist_path(-1, 0.0) + [site("thirdmc", (0.012, 0, -0.010))], 80.0)
    # pronators / supinator-like: wrap around the forearm axis is approximated with via points
    muscle("PT", [site("ulna", (0.03, 0.022, -0.004)), site("ulna", (0.07, 0.010, -0.020)), site("radius", (0.11, -0.018, -0.004))], 300.0)
    muscle("PQ", [site("ulna", (0.21, 0.020, -0.006)), site("ulna", (0.213, 0.0, -0.019)), site("radius", (0.216, -0.020, -0.004))], 150.0)
    muscle("EIP", [site("ulna", (0.15, 0.010, 0.012))] + wrist_path(+1, -0.016) + finger_path(2, +1, 3), 70.0)
    # --- thumb
    def th(body, along, off_abd, off_flex):
        base = {"firstmc": 0.0, "proximal_thumb": 0.0, "distal_thumb": 0.0}[body]
        return site(body, (base + along) * tdir + off_abd * tabd + off_flex * tflex)
    dors = np.cross(tdir, tflex)  # thumb dorsal direction (== -tabd up to sign)
    def thd(body, along, h):  # h>0 dorsal (extensor), h<0 palmar (flexor) w.r.t. flexion axis
        return site(body, along * tdir + h * np.cross(-tflex, tdir))
    muscle("EPL", [site("ulna", (0.10, 0.012, 0.012))] + wrist_path(+1, -0.014) +
           [thd("firstmc", 0.012, 0.008), thd("firstmc", TL[0] - 0.008, 0.007),
            ("sphere", "thmp_wrap", "thmp_side"), thd("proximal_thumb", 0.008, 0.006),
            thd("proximal_thumb", TL[1] - 0.006, 0.005), thd("distal_thumb", 0.006, 0.004)], 90.0)
    muscle("EPB", [site("radius", (0.16, -0.010, 0.012)), site("radius", (WX - 0.01, -0.024, 0.006)),
                   thd("firstmc", 0.010, 0.008), thd("firstmc", TL[0] - 0.008, 0.007),
                   thd("proximal_thumb", 0.008, 0.006)], 60.0)
    muscle("FPL", [site("radius", (0.10, -0.008, -0.012))] + wrist_path(-1, -0.018) +
           [thd("firstmc", 0.012, -0.009), thd("firstmc", TL[0] - 0.008, -0.008),
            thd("proximal_thumb", 0.008, -0.007), thd("proximal_thumb", TL[1] - 0.006, -0.006),
            thd("distal_thumb", 0.007, -0.005)], 120.0)
    muscle("APL", [site("radius", (0.14, -0.006, 0.012)), site("radius", (WX - 0.008, -0.026, 0.0)),
                   site("firstmc", 0.008 * tdir + 0.009 * tabd)], 100.0)
    muscle("OP", [site("capitate", (0.004, -0.004, -0.012)), site("firstmc", 0.030 * tdir - 0.008 * tabd - 0.004 * np.cross(-tflex, tdir))], 80.0)
    # --- extrinsic finger flexors / extensors
    for k in (2, 3, 4, 5):
        muscle(f"FDS{k}", [site("ulna", (0.04 + 0.01 * k, fy[k] * 0.5, -0.014))] + wrist_path(-1, fy[k] * 0.6) +
               finger_path(k, -1, 2), {2: 75.0, 3: 85.0, 4: 65.0, 5: 45.0}[k])
    for k in (2, 3, 4, 5):
        muscle(f"FDP{k}", [site("ulna", (0.05 + 0.01 * k, fy[k] * 0.5, -0.010))] + wrist_path(-1, fy[k] * 0.5) +
               finger_path(k, -1, 3), {2: 80.0, 3: 90.0, 4: 75.0, 5: 55.0}[k])
    for k in (2, 3, 4, 5):
        muscle(f"EDC{k}", [site("ulna", (0.06 + 0.01 * k, fy[k] * 0.5, 0.012))] + wrist_path(+1, fy[k] * 0.55) +
               finger_path(k, +1, 3), {2: 150.0, 3: 165.0, 4: 140.0, 5: 100.0}[k])
    muscle("EDM", [site("ulna", (0.12, 0.018, 0.011))] + wrist_path(+1, 0.020) + finger_path(5, +1, 3), 60.0)
    # --- intrinsics: metacarpal -> proximal phalanx sides (ab/adduction + MCP flexion)
    for k in (2, 3, 4, 5):
        muscle(f"RI{k}", [site(mcname[k], (0.020, -0.007, 0.0)), site(mcname[k], (fmc[k] - 0.010, -0.008, -0.003)),
                          site(f"proxph{k}", (0.010, -0.007, -0.002))], 60.0)
    for k in (2, 3, 4, 5):
        muscle(f"LU_RB{k}", [site(mcname[k], (0.030, -0.004, -0.008)), site(mcname[k], (fmc[k] - 0.010, -0.005, -0.009)),
                             site(f"proxph{k}", (0.012, -0.004, -0.006)), site(f"proxph{k}", (flen[k][0] - 0.008, -0.003, 0.004)),
                             site(f"midph{k}", (0.006, 0.0, 0.005))], 25.0)
    for k in (2, 3, 4, 5):
        muscle(f"UI_UB{k}", [site(mcname[k], (0.020, 0.007, 0.0)), site(mcname[k], (fmc[k] - 0.010, 0.008, -0.003)),
                             site(f"proxph{k}", (0.010, 0.007, -0.002))], 50.0)
    assert [a.name for a in s.actuators] == HAND_MUSCLES
    assert [j.name for j in s.joints] == HAND_JOINTS
    if self_collision:
        _add_hand_capsules(s)
        s.nconmax = 10
        pairs = []
        for k in (2, 3, 4):                       # neighbouring fingers, segment against segment
            for seg in ("proxph", "midph", "distph"):
                pairs.append((f"col_{seg}{k}", f"col_{seg}{k + 1}"))
        pairs += [("col_distal_thumb", "col_proxph2"), ("col_distal_thumb", "col_midph2"), ("col_distal_thumb", "col_distph2"),
                  ("col_distal_thumb", "col_midph3"), ("col_distal_thumb", "col_distph3"), ("col_proximal_thumb", "col_proxph2"),
                  ("col_distal_thumb", "col_mc2")]
        pairs += [(f"col_distph{k}", f"col_mc{k}") for k in (2, 3, 4, 5)]     # a fully curled finger reaches its own metacarpal
        for g1, g2 in pairs:
            s.add_contact_pair(g1, g2, condim=3, friction=(1.0, 0.005, 0.0001))
    return s


# ----------------------------------------------------------------------------- legs
LEG_JOINTS_SIDE = ["hip_flexion", "hip_adduction", "hip_rotation", "knee_angle_translation2", "knee_angle_translation1",
                   "knee_angle", "knee_angle_rotation2", "knee_angle_rotation3", "ankle_angle", "subtalar_angle",
                   "mtp_angle", "knee_angle_beta_translation2", "knee_angle_beta_translation1",
                   "knee_angle_beta_rotation1"]
LEG_MUSCLES_SIDE = ["addbrev", "addlong", "addmagDist", "addmagIsch", "addmagMid", "addmagProx", "bflh", "bfsh", "edl",
                    "ehl", "fdl", "fhl", "gaslat", "gasmed", "glmax1", "glmax2", "glmax3", "glmed1", "glmed2",
                    "glmed3", "glmin1", "glmin2", "glmin3", "grac", "iliacus", "perbrev", "perlong", "piri", "psoas",
                    "recfem", "sart", "semimem", "semiten", "soleus", "tfl", "tibant", "tibpost", "vasint", "vaslat",
                    "vasmed"]


def _jname(base, side):
    """MyoLeg naming: knee_angle_r_translation2, hip_flexion_r, ..."""
    if base.startswith("knee_angle"):
        return "knee_angle_" + side + base[len("knee_angle"):]
    return base + "_" + side


# knee coupling polynomials q = poly(knee_angle) (synthetic magnitudes, same structure as MyoLeg's 7 joint equalities
# per knee): tibia translations / secondary rotations, patella translations / rotation
_KNEE_POLY = {"knee_angle_translation2": (0.0, -0.004, 0.0015), "knee_angle_translation1": (0.0, 0.003, -0.001),
              "knee_angle_rotation2": (0.0, 0.04, -0.01), "knee_angle_rotation3": (0.0, 0.09, -0.025),
              "knee_angle_beta_translation2": (0.0, -0.018, 0.002), "knee_angle_beta_translation1": (0.0, -0.012, 0.0),
              "knee_angle_beta_rotation1": (0.0, 0.75, -0.05)}



def _qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _reframe_z(s: ModelSpec, angle: float):
    """Re-express every body-local quantity of `s` in body frames rotated by `angle` about z (authoring aid:
    the leg is authored with y forward and handed over with MyoLeg's x-forward frames)."""
    r = np.array([math.cos(angle / 2), 0.0, 0.0, math.sin(angle / 2)]); rc = r * np.array([1, -1, -1, -1])
    c, sn = math.cos(angle), math.sin(angle)
    R = np.array([[c, -sn, 0], [sn, c, 0], [0, 0, 1.0]])
    for b in s.bodies[1:]:
        b.pos = R @ b.pos; b.ipos = R @ b.ipos
        b.quat = _qmul(_qmul(r, b.quat), rc); b.iquat = _qmul(r, b.iquat)
    for j in s.joints:
        j.pos = R @ j.pos; j.axis = R @ j.axis
    s.sites = [(n, b, R @ p) for (n, b, p) in s.sites]
    for g in s.geoms:
        g["pos"] = R @ g["pos"]; g["quat"] = _qmul(r, g["quat"])


def make_leg(implicit: bool = False) -> ModelSpec:
    """``implicit``: the MuJoCo-default version of the model -- force-velocity range vmax = 1.5 L0/s on every muscle, no
    reflected-inertia padding on the foot joints, integrator implicitfast (what explicit Euler cannot step at this timestep:
    see DESIGN.md).  The default (explicit Euler) version softens both.

    myoLeg: free-floating pelvis + torso, 2 x 14 leg joints (34 DoF, nq 35), 80 muscles, 14 knee joint-equalities,
    8 foot-ground contact pairs.  Authored below with x right / y forward / z up, then re-framed to MyoLeg's body frames
    (x forward, y left, z up).  The keyframes carry the root quaternion (0.7071, 0, 0, -0.7071), i.e. the model faces
    world -y: that is what makes the reference's reward / termination arithmetic consistent (walk_v0.py:438-446 negates
    cvel and rewards y-velocity 1.2; walk_v0.py:461-472 terminates when |R[0,0]| of the root exceeds max_rot).
    Joint / muscle names and dimensions: SURVEY.md 8d (walk_v0.py:236-241,438-451; docs/source/suite.rst)."""
    s = ModelSpec("myolegs_implicitfast" if implicit else "myolegs", timestep=0.001, integrator=3 if implicit else 0)  # x frame_skip 10 (BaseV0 default) = 0.01 s per env step: hip_period 100 -> 1 s stride
    s.add_geom("floor", "world", "plane", (0, 0, 0))
    PZ = 0.982
    s.add_body("pelvis", "world", pos=(0, 0, PZ), mass=11.8, ipos=(0, -0.04, 0.0), inertia=(0.10, 0.09, 0.06))
    s.add_joint("root", "pelvis", "free")
    s.add_body("torso", "pelvis", pos=(0, -0.03, 0.08), mass=34.0, ipos=(0, -0.01, 0.27), inertia=(1.47, 1.43, 0.76))
    s.add_site("pelvis_mark", "pelvis", (0, 0, 0))
    s.add_site("pelvis", "pelvis", (0, 0, 0))            # tip / target pair of myoLegStandRandom-v0 (myobase/__init__.py:434)
    s.add_site("pelvis_target", "world", (0, 0, PZ))
    for side, sx in (("r", 1.0), ("l", -1.0)):
        def X(p):
            return (sx * p[0], p[1], p[2])
        B = {k: f"{k}_{side}" for k in ("femur", "tibia", "talus", "calcn", "toes", "patella")}
        B["pelvis"] = "pelvis"
        hinge = dict(damping=0.1, armature=0.005)
        s.add_body(B["femur"], "pelvis", pos=X((0.085, 0.0, -0.07)), mass=9.3, ipos=(0, 0, -0.17),
                   inertia=(0.134, 0.134, 0.035))
        s.add_joint(_jname("hip_flexion", side), B["femur"], "hinge", axis=(1, 0, 0), range=(-0.52, 2.09), **hinge)
        s.add_joint(_jname("hip_adduction", side), B["femur"], "hinge", axis=(0, sx, 0), range=(-0.87, 0.52), **hinge)
        s.add_joint(_jname("hip_rotation", side), B["femur"], "hinge", axis=(0, 0, sx), range=(-0.70, 0.70), **hinge)
        s.add_body(B["tibia"], B["femur"], pos=(0, 0, -0.41), mass=3.7, ipos=(0, 0, -0.187), inertia=(0.05, 0.05, 0.005))
        minor = dict(damping=0.5, armature=0.01)
        s.add_joint(_jname("knee_angle_translation2", side), B["tibia"], "slide", axis=(0, 0, 1), **minor)
        s.add_joint(_jname("knee_angle_translation1", side), B["tibia"], "slide", axis=(0, 1, 0), **minor)
        s.add_joint(_jname("knee_angle", side), B["tibia"], "hinge", axis=(-1, 0, 0), range=(0.0, 2.09), **hinge)
        s.add_joint(_jname("knee_angle_rotation2", side), B["tibia"], "hinge", axis=(0, sx, 0), damping=0.1, armature=0.002)
        s.add_joint(_jname("knee_angle_rotation3", side), B["tibia"], "hinge", axis=(0, 0, sx), damping=0.1, armature=0.002)
        s.add_body(B["talus"], B["tibia"], pos=(0, 0, -0.43), mass=0.1, inertia=(0.001, 0.001, 0.001))
        s.add_joint(_jname("ankle_angle", side), B["talus"], "hinge", axis=(1, 0, 0), range=(-0.70, 0.52),
                    damping=0.5, armature=0.003 if implicit else 0.03)
        s.add_body(B["calcn"], B["talus"], pos=X((-0.005, -0.049, -0.042)), mass=1.25, ipos=(0, 0.09, 0.012),
                   inertia=(0.004, 0.0014, 0.0041))
        ax = np.array([sx * 0.12, 0.787, 0.605]); ax /= np.linalg.norm(ax)      # oblique subtalar axis (inversion +)
        s.add_joint(_jname("subtalar_angle", side), B["calcn"], "hinge", axis=tuple(-ax if sx > 0 else ax),
                    range=(-0.35, 0.35), damping=0.3, armature=0.002 if implicit else 0.015)
        s.add_body(B["toes"], B["calcn"], pos=(0, 0.179, -0.002), mass=0.22, ipos=(0, 0.03, -0.005),
                   inertia=(0.0001, 0.0002, 0.0002))
        s.add_joint(_jname("mtp_angle", side), B["toes"], "hinge", axis=(1, 0, 0), range=(-0.52, 0.52),
                    damping=0.1, armature=0.003, stiffness=2.0)
        s.add_body(B["patella"], B["femur"], pos=(0, 0.045, -0.40), mass=0.09, inertia=(1e-4, 1e-4, 1e-4))
        s.add_joint(_jname("knee_angle_beta_translation2", side), B["patella"], "slide", axis=(0, 0, 1), **minor)
        s.add_joint(_jname("knee_angle_beta_translation1", side), B["patella"], "slide", axis=(0, 1, 0), **minor)
        s.add_joint(_jname("knee_angle_beta_rotation1", side), B["patella"], "hinge", pos=(0, -0.045, -0.01),
                    axis=(-1, 0, 0), damping=0.1, armature=0.002)
        for base, pc in _KNEE_POLY.items():
            s.add_equality_joint(_jname(base, side), _jname("knee_angle", side), pc)

        # ---- foot contact geometry (four spheres per foot) against the floor plane
        s.add_geom(f"heel_{side}", B["calcn"], "sphere", (0.025,), pos=X((0.0, 0.012, -0.005)))
        s.add_geom(f"ball_lat_{side}", B["calcn"], "sphere", (0.02,), pos=X((0.035, 0.155, -0.01)))
        s.add_geom(f"ball_med_{side}", B["calcn"], "sphere", (0.02,), pos=X((-0.025, 0.165, -0.01)))
        s.add_geom(f"toe_{side}", B["toes"], "sphere", (0.016,), pos=X((0.0, 0.035, -0.012)))
        for gname in (f"heel_{side}", f"ball_lat_{side}", f"ball_med_{side}", f"toe_{side}"):
            s.add_contact_pair("floor", gname, condim=3, friction=(1.0, 0.005, 0.0001))

        # ---- knee-extensor wrap cylinder on the femoral condyles (axis along x), anterior side site
        s.add_geom(f"knee_wrap_{side}", B["femur"], "cylinder", size=(0.04, 0.06), pos=(0, 0, -0.41),
                   quat=(math.cos(math.pi / 4), 0.0, math.sin(math.pi / 4), 0.0))
        s.add_site(f"knee_side_{side}", B["femur"], (0, 0.09, -0.41))

        cnt = [0]

        def site(body, p):
            nm = f"ls_{side}_{cnt[0]}"; cnt[0] += 1
            s.add_site(nm, B[body], X(p))
            return ("site", nm)

        KW = ("cylinder", f"knee_wrap_{side}", f"knee_side_{side}")

        def muscle(name, path, force):
            tn = f"{name}_{side}_tendon"
            s.add_tendon(tn, path)
            # vmax 10 L0/s (physiological): with MuJoCo's default 1.5 the force-velocity slope of the big ankle /
            # knee muscles is too stiff for explicit Euler at this timestep (synthetic models)
            s.add_muscle(f"{name}_{side}", tn, force=force, range=(0.60, 1.35), vmax=1.5 if implicit else 10.0)

        P, F, T, Cn, TO = "pelvis", "femur", "tibia", "calcn", "toes"
        muscle("addbrev", [site(P, (0.020, 0.010, -0.090)), site(F, (0.005, -0.005, -0.13))], 600.0)
        muscle("addlong", [site(P, (0.020, 0.020, -0.085)), site(F, (0.005, -0.003, -0.21))], 900.0)
        muscle("addmagDist", [site(P, (0.030, -0.060, -0.110)), site(F, (0.005, -0.005, -0.23))], 600.0)
        muscle("addmagIsch", [site(P, (0.035, -0.070, -0.115)), site(F, (-0.010, 0.000, -0.39))], 600.0)
        muscle("addmagMid", [site(P, (0.030, -0.050, -0.105)), site(F, (0.005, -0.005, -0.17))], 600.0)
        muscle("addmagProx", [site(P, (0.025, -0.040, -0.100)), site(F, (0.005, -0.005, -0.11))], 600.0)
        muscle("bflh", [site(P, (0.060, -0.080, -0.100)), site(T, (0.035, -0.020, -0.04))], 1300.0)
        muscle("bfsh", [site(F, (0.010, -0.005, -0.220)), site(T, (0.035, -0.020, -0.04))], 600.0)
        muscle("edl", [site(T, (0.020, 0.020, -0.12)), site(T, (0.010, 0.035, -0.40)), site(Cn, (0.005, 0.110, 0.030)),
                       site(TO, (0.010, 0.040, 0.005))], 350.0)
        muscle("ehl", [site(T, (0.010, 0.020, -0.20)), site(T, (0.000, 0.035, -0.40)), site(Cn, (-0.010, 0.120, 0.030)),
                       site(TO, (-0.015, 0.045, 0.008))], 160.0)
        muscle("fdl", [site(T, (-0.015, -0.015, -0.20)), site(T, (-0.020, -0.020, -0.42)), site(Cn, (-0.020, 0.050, 0.000)),
                       site(TO, (0.000, 0.030, -0.008))], 300.0)
        muscle("fhl", [site(T, (0.000, -0.020, -0.25)), site(T, (-0.015, -0.030, -0.42)), site(Cn, (-0.020, 0.060, -0.005)),
                       site(TO, (-0.015, 0.040, -0.008))], 350.0)
        muscle("gaslat", [site(F, (0.020, -0.015, -0.395)), site(Cn, (0.003, -0.010, 0.020))], 1100.0)
        muscle("gasmed", [site(F, (-0.020, -0.015, -0.395)), site(Cn, (-0.003, -0.010, 0.020))], 1600.0)
        muscle("glmax1", [site(P, (0.050, -0.080, 0.040)), site(F, (0.040, -0.020, -0.06))], 700.0)
        muscle("glmax2", [site(P, (0.045, -0.090, 0.000)), site(F, (0.035, -0.020, -0.10))], 900.0)
        muscle("glmax3", [site(P, (0.030, -0.100, -0.050)), site(F, (0.030, -0.015, -0.14))], 700.0)
        muscle("glmed1", [site(P, (0.100, 0.030, 0.090)), site(F, (0.055, -0.003, -0.005))], 900.0)
        muscle("glmed2", [site(P, (0.110, -0.010, 0.090)), site(F, (0.056, -0.006, -0.005))], 600.0)
        muscle("glmed3", [site(P, (0.090, -0.050, 0.070)), site(F, (0.055, -0.010, -0.005))], 700.0)
        muscle("glmin1", [site(P, (0.100, 0.020, 0.040)), site(F, (0.050, 0.005, -0.005))], 300.0)
        muscle("glmin2", [site(P, (0.105, 0.000, 0.040)), site(F, (0.050, 0.002, -0.005))], 300.0)
        muscle("glmin3", [site(P, (0.095, -0.020, 0.035)), site(F, (0.050, -0.002, -0.005))], 300.0)
        muscle("grac", [site(P, (0.015, 0.000, -0.100)), site(T, (-0.020, 0.015, -0.06))], 250.0)
        muscle("iliacus", [site(P, (0.060, 0.040, 0.050)), site(P, (0.075, 0.045, -0.060)), site(F, (0.000, -0.005, -0.07))], 1100.0)
        muscle("perbrev", [site(T, (0.030, -0.010, -0.25)), site(T, (0.030, -0.030, -0.42)), site(Cn, (0.030, 0.060, 0.000))], 500.0)
        muscle("perlong", [site(T, (0.030, -0.005, -0.15)), site(T, (0.032, -0.030, -0.42)), site(Cn, (0.025, 0.070, -0.010))], 900.0)
        muscle("piri", [site(P, (0.020, -0.100, 0.020)), site(F, (0.050, -0.010, 0.005))], 500.0)
        muscle("psoas", [site(P, (0.030, 0.030, 0.120)), site(P, (0.070, 0.050, -0.060)), site(F, (-0.005, -0.008, -0.06))], 1400.0)
        muscle("recfem", [site(P, (0.080, 0.040, -0.030)), KW, site(T, (0.000, 0.035, -0.06))], 2200.0)
        muscle("sart", [site(P, (0.090, 0.050, 0.000)), site(F, (-0.030, 0.000, -0.36)), site(T, (-0.020, 0.020, -0.06))], 250.0)
        muscle("semimem", [site(P, (0.050, -0.080, -0.105)), site(T, (-0.030, -0.020, -0.05))], 1100.0)
        muscle("semiten", [site(P, (0.050, -0.085, -0.110)), site(T, (-0.030, -0.015, -0.07))], 600.0)
        muscle("soleus", [site(T, (0.000, -0.020, -0.10)), site(Cn, (0.000, -0.010, 0.020))], 3600.0)
        muscle("tfl", [site(P, (0.110, 0.040, 0.050)), site(F, (0.050, 0.010, -0.10)), site(T, (0.040, 0.010, -0.04))], 450.0)
        muscle("tibant", [site(T, (0.015, 0.020, -0.15)), site(T, (0.000, 0.035, -0.41)), site(Cn, (-0.015, 0.090, 0.025))], 1100.0)
        muscle("tibpost", [site(T, (-0.005, -0.015, -0.15)), site(T, (-0.020, -0.025, -0.42)), site(Cn, (-0.025, 0.050, 0.005))], 1400.0)
        muscle("vasint", [site(F, (0.010, 0.025, -0.20)), KW, site(T, (0.000, 0.035, -0.06))], 1700.0)
        muscle("vaslat", [site(F, (0.030, 0.015, -0.22)), KW, site(T, (0.005, 0.035, -0.06))], 3500.0)
        muscle("vasmed", [site(F, (-0.020, 0.015, -0.25)), KW, site(T, (-0.005, 0.035, -0.06))], 2300.0)

    _reframe_z(s, -math.pi / 2)      # authored (x right, y forward)  ->  handed over (x forward, y left)
    names = ["root"] + [_jname(b, sd) for sd in ("r", "l") for b in LEG_JOINTS_SIDE]
    assert [j.name for j in s.joints] == names
    assert [a.name for a in s.actuators] == [f"{m}_{sd}" for sd in ("r", "l") for m in LEG_MUSCLES_SIDE]

    # ---- keyframes: the independent coordinates (hip x3, knee, ankle, subtalar, mtp) and their velocities are the
    # reference's own numbers (myosuite/envs/myo/assets/leg/myolegs_chasetag.xml:53-56; walk_v0.py:271,334-363 uses
    # key 0 = stand, key 2 / 3 = mid-stride); the 7 coupled knee / patella coordinates follow OUR coupling polynomials
    # and the root height is re-grounded on the synthetic feet (_ground_keyframes).
    IND = [0, 1, 2, 5, 8, 9, 10]     # hip_flexion, hip_adduction, hip_rotation, knee_angle, ankle, subtalar, mtp

    def key(qr, ql, vr=None, vl=None, vy=0.0):
        q = np.zeros(35); q[2] = PZ; q[3] = 0.707388; q[6] = -0.706825        # facing world -y
        q[3:7] /= np.linalg.norm(q[3:7])
        v = np.zeros(34); v[1] = vy
        for k, (qq, vv) in enumerate(((qr, vr), (ql, vl))):
            o = 7 + 14 * k
            for i, val in zip(IND, qq):
                q[o + i] = val
            knee = q[o + 5]
            for base, pc in _KNEE_POLY.items():
                q[o + LEG_JOINTS_SIDE.index(base)] = sum(c * knee ** i for i, c in enumerate(pc))
            if vv is not None:
                for i, val in zip(IND, vv):
                    v[6 + 14 * k + i] = val
                dknee = v[6 + 14 * k + 5]
                for base, pc in _KNEE_POLY.items():   # velocities consistent with the couplings
                    v[6 + 14 * k + LEG_JOINTS_SIDE.index(base)] = dknee * sum(i * c * knee ** (i - 1) for i, c in enumerate(pc) if i > 0)
        return q, v
    stand = key((0.161153, -0.0279385, -0.041886, 0.461137, 0.334, -0.00117055, -0.000125295),) * 2 if False else \
        key((0.161153, -0.0279385, -0.041886, 0.461137, 0.334, -0.00117055, -0.000125295),
            (0.161153, -0.0279385, -0.041886, 0.461137, 0.334, -0.00117055, -0.000125295))
    crouch = key((0.405648, -0.020957, -0.118677, 0.7329, 0.40143, -0.006982, -0.02618),
                 (0.405648, -0.020957, -0.118677, 0.7329, 0.40143, -0.006982, -0.02618))
    swing = (-0.2326, -0.0279385, -0.041886, 1.227, 0.1672, -0.00117055, -0.000125295)
    stance = (-0.1652, -0.0279385, -0.041886, 0.0888, -0.019, -0.00117055, -0.000125295)
    v_swing = (4.9066, 0.0, 0.0, -3.597, 0.633, 0.0, 0.0)
    strideR = key(swing, stance, v_swing, (0.175, 0.0, 0.0, 0.175, 0.988, 0.0, 0.0), vy=-1.5)
    strideL = key(stance, swing, (-0.576, 0.0, 0.0, 0.175, 0.988, 0.0, 0.0), v_swing, vy=-1.5)
    s.keys = [stand, crouch, strideR, strideL]
    return s


# ----------------------------------------------------------------------------- hand + object (reorient)
# capsule sizes [radius, half-length, (unused)] of the reference's reset tables -- data, not code:
# myosuite/envs/myo/myobase/reorient_sar_v0.py:207-210 (Geometries8EnvV0) and :322-348 (Geometries100EnvV0)
REORIENT_CAPS_8 = [(0.013, 0.025, 0.025), (0.019, 0.040, 0.040)]
REORIENT_CAPS_100 = [(0.0162, 0.0422, 0.0484), (0.016, 0.0457, 0.0496), (0.0187, 0.0259, 0.0248), (0.0192, 0.0483, 0.0216),
                     (0.0213, 0.0218, 0.0481), (0.0169, 0.0331, 0.0388), (0.0138, 0.0299, 0.0471), (0.0194, 0.0252, 0.0419),
                     (0.014, 0.0362, 0.0201), (0.0125, 0.029, 0.0298), (0.0162, 0.0396, 0.0323), (0.019, 0.0365, 0.0421),
                     (0.0143, 0.0228, 0.0255), (0.0147, 0.0391, 0.0369), (0.0192, 0.0324, 0.043), (0.0145, 0.0491, 0.0234),
                     (0.013, 0.0458, 0.0457), (0.0187, 0.0219, 0.0434), (0.0198, 0.0276, 0.0238), (0.0175, 0.0375, 0.0339),
                     (0.0191, 0.049, 0.0472), (0.0145, 0.0425, 0.0356), (0.0134, 0.0291, 0.0379), (0.0185, 0.0445, 0.0454),
                     (0.0164, 0.041, 0.0328)]


# ellipsoid / cylinder / box tables of the same reset code (reorient_sar_v0.py:179-205 and :267-377)
REORIENT_ELLIPS_8 = [(0.011, 0.025, 0.025), (0.019, 0.04, 0.04)]
REORIENT_CYL_8 = [(0.013, 0.025, 0.025), (0.019, 0.04, 0.04)]
REORIENT_BOX_8 = [(0.017, 0.017, 0.017), (0.023, 0.023, 0.023)]
REORIENT_ELLIPS_100 = [(0.02843, 0.0256, 0.02902), (0.01057, 0.02655, 0.0328), (0.01126, 0.0273, 0.04264), (0.02641, 0.03524, 0.02831), (0.02804, 0.03722, 0.04313), (0.02305, 0.04456, 0.03709), (0.02332, 0.02673, 0.02606), (0.01247, 0.03233, 0.03759), (0.02199, 0.029, 0.04484), (0.02674, 0.0428, 0.03764), (0.02278, 0.04006, 0.03556), (0.02392, 0.04095, 0.03467), (0.01928, 0.0348, 0.03044), (0.02388, 0.03644, 0.02817), (0.02739, 0.04338, 0.03457), (0.00962, 0.04047, 0.02614), (0.0163, 0.04443, 0.04326), (0.02417, 0.03157, 0.04038), (0.01927, 0.02814, 0.03786), (0.02477, 0.04456, 0.04493), (0.01656, 0.0291, 0.03996), (0.01763, 0.03877, 0.03636), (0.01915, 0.0346, 0.04245), (0.02485, 0.03324, 0.02881), (0.00856, 0.04185, 0.03749)]
REORIENT_BOX_100 = [(0.02295, 0.02306, 0.02221), (0.02447, 0.0185, 0.02192), (0.01853, 0.01837, 0.01546), (0.01586, 0.02079, 0.022), (0.02293, 0.02116, 0.02255), (0.01542, 0.01651, 0.02381), (0.0186, 0.02402, 0.02333), (0.01782, 0.01584, 0.02208), (0.01907, 0.0195, 0.02161), (0.01751, 0.0211, 0.01864), (0.02258, 0.02334, 0.01856), (0.02195, 0.01617, 0.02438), (0.01627, 0.02254, 0.02073), (0.02364, 0.01946, 0.01777), (0.01754, 0.02463, 0.01549), (0.02394, 0.02382, 0.02387), (0.01997, 0.02372, 0.02032), (0.01741, 0.02316, 0.02203), (0.02032, 0.0217, 0.02432), (0.01961, 0.0248, 0.0176), (0.01906, 0.01999, 0.02399), (0.02472, 0.01826, 0.0151), (0.01636, 0.0158, 0.01958), (0.01542, 0.02434, 0.02237), (0.01731, 0.02185, 0.02019)]
REORIENT_CYL_100 = [(0.0118, 0.044, 0.0265), (0.0189, 0.0316, 0.0415), (0.0123, 0.0364, 0.0238), (0.0145, 0.0362, 0.032), (0.0146, 0.0306, 0.0283), (0.0155, 0.0237, 0.0383), (0.0198, 0.0323, 0.03), (0.011, 0.0368, 0.0343), (0.0197, 0.0305, 0.0206), (0.0162, 0.0277, 0.0329), (0.0202, 0.0339, 0.0223), (0.0181, 0.0204, 0.0314), (0.0118, 0.0374, 0.0228), (0.0197, 0.0319, 0.0375), (0.0133, 0.0392, 0.0448), (0.0157, 0.0236, 0.0301), (0.0178, 0.0351, 0.0414), (0.0123, 0.024, 0.0399), (0.0158, 0.0254, 0.022), (0.0192, 0.021, 0.0355), (0.011, 0.0266, 0.0261), (0.0121, 0.0369, 0.0226), (0.012, 0.0221, 0.0254), (0.0213, 0.0298, 0.0297), (0.0164, 0.0302, 0.025)]


def reorient_tables(geometries: str):
    """[4][ntab][3] size tables in geom-type order capsule(3), ellipsoid(4), cylinder(5), box(6)"""
    if str(geometries) in ("ID", "OOD"):      # 4 x 250 rows of the reference's test envs (tools/extract_reorient_tables.py)
        import os
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "envs", "data", "reorient_tables.npz")
        return np.load(path)[str(geometries)].astype(np.float32)
    if str(geometries) == "8":
        return np.array([REORIENT_CAPS_8, REORIENT_ELLIPS_8, REORIENT_CYL_8, REORIENT_BOX_8], np.float32)
    return np.array([REORIENT_CAPS_100, REORIENT_ELLIPS_100, REORIENT_CYL_100, REORIENT_BOX_100], np.float32)


def _quat_z_to(v):
    """quaternion rotating the local z axis onto unit vector v"""
    v = np.asarray(v, np.float64) / np.linalg.norm(v)
    ax = np.cross([0.0, 0.0, 1.0], v); sn = np.linalg.norm(ax); cs = v[2]
    if sn < 1e-12:
        return (1.0, 0.0, 0.0, 0.0) if cs > 0 else (0.0, 1.0, 0.0, 0.0)
    ang = math.atan2(sn, cs); ax = ax / sn
    return (math.cos(ang / 2),) + tuple(math.sin(ang / 2) * ax)


def _hand_palm_up_with_capsules(name: str):
    """make_hand() mounted so that pro_sup = -1.5 turns the palm up, plus collision capsules along metacarpals, phalanges and
    the carpal row.  Returns (spec, capsule geom names)."""
    s = make_hand()
    s.name = name
    s.nconmax = 8
    phi = -(math.pi - 1.5)                         # base roll: pro_sup = -1.5 then sums to -pi, palm (local -z) up
    s.bodies[s._bname["ulna"]].quat = np.array([math.cos(phi / 2), math.sin(phi / 2), 0.0, 0.0])
    return s, _add_hand_capsules(s)


def _add_hand_capsules(s: ModelSpec, only=None):
    """Collision capsules along metacarpals, phalanges and the carpal row of make_hand() (``only``: restrict to these bodies).
    Returns the capsule geom names."""
    caps = []

    def cap(name, body, r, p0, p1):
        if only is not None and body not in only:
            return
        p0, p1 = np.asarray(p0, float), np.asarray(p1, float)
        s.add_geom(name, body, "capsule", (r, 0.5 * np.linalg.norm(p1 - p0)), pos=tuple(0.5 * (p0 + p1)), quat=_quat_z_to(p1 - p0))
        caps.append(name)

    fmc = {2: 0.066, 3: 0.064, 4: 0.058, 5: 0.053}
    flen = {2: (0.043, 0.025, 0.019), 3: (0.047, 0.029, 0.020), 4: (0.043, 0.027, 0.020), 5: (0.034, 0.020, 0.018)}
    mcname = {2: "secondmc", 3: "thirdmc", 4: "fourthmc", 5: "fifthmc"}
    for k in (2, 3, 4, 5):
        cap(f"col_mc{k}", mcname[k], 0.009, (0.006, 0, 0), (fmc[k] - 0.006, 0, 0))
        for body, L, r in ((f"proxph{k}", flen[k][0], 0.008), (f"midph{k}", flen[k][1], 0.007), (f"distph{k}", flen[k][2], 0.006)):
            cap(f"col_{body}", body, r, (0.004, 0, 0), (L - 0.003, 0, 0))
    tdir = np.array([0.55, -0.70, -0.46]); tdir /= np.linalg.norm(tdir)
    for body, L, r in (("firstmc", 0.044, 0.009), ("proximal_thumb", 0.032, 0.008), ("distal_thumb", 0.026, 0.007)):
        cap(f"col_{body}", body, r, 0.004 * tdir, (L - 0.003) * tdir)
    cap("col_carpal", "capitate", 0.012, (0.008, -0.026, 0), (0.008, 0.028, 0))
    return caps


def _make_hand_with_object(kind: str) -> ModelSpec:
    """myoHand + free-moving object (3 slide + 3 hinge joints, NOT a free joint) + static target, the structure of
    myosuite/envs/myo/assets/hand/myohand_sar.xml:22-57: nq = nv = 29, 39 muscles, obs 200.  The forearm is mounted so that
    init_qpos[0] = pro_sup = -1.5 (reorient_sar_v0.py:113-114) turns the palm up; collision capsules along metacarpals,
    phalanges and the carpal row catch the object.  Object geom: compiled as a capsule; type (capsule / ellipsoid /
    cylinder / box) and size are per-env model deltas re-drawn every episode from the reference's tables."""
    s, caps = _hand_palm_up_with_capsules("myohand_sar" if kind == "reorient" else "myohand_pen")
    # Contact and row bounds (mjModel.nconmax / njmax, independent as in MuJoCo; both are honoured by oracle and kernel, surplus is
    # dropped and flagged): twelve contacts -- a box object makes up to two per finger capsule (mjc_CapsuleBox), and under random
    # actions the pen / box envs do reach 9...11 -- within the row bound the models always had (23 limit rows + 8 x 4 = 55, rounded to
    # 56): an env's efc_J table is efc_rows x 36 words of LDS, and 64 rows cost the 32-wide kernel its second resident wave per SIMD
    # (measured on reorient: 4.10 -> 2.46 M env-steps/s).
    s.nconmax = 12
    s.njmax = 56
    # object above the palm centre (world frame at init: local (x, y, z) -> (x, -y, 1 - z) for the rolled forearm)
    OX, OZ = 0.325, 1.0 + 0.009 + 0.022
    eul = 1.27                                                              # myohand_sar.xml:26  euler="0 1.27 0"
    oq = (math.cos(eul / 2), 0.0, math.sin(eul / 2), 0.0)
    m0 = 4.0 / 3.0 * math.pi * 0.015 * 0.015 * 0.045 * 1500.0              # ellipsoid .015 .015 .045, density 1500 (xml:34)
    s.add_body("Object", "world", pos=(OX, 0.004, OZ), quat=oq, mass=1.2,   # body_mass = 1.2 at every reset (py:417)
               inertia=(m0 / 5 * (0.015 ** 2 + 0.045 ** 2), m0 / 5 * (0.015 ** 2 + 0.045 ** 2), m0 / 5 * 2 * 0.015 ** 2))
    for nm, typ, ax in (("OBJTx", "slide", (1, 0, 0)), ("OBJTy", "slide", (0, 1, 0)), ("OBJTz", "slide", (0, 0, 1)),
                        ("OBJRx", "hinge", (1, 0, 0)), ("OBJRy", "hinge", (0, 1, 0)), ("OBJRz", "hinge", (0, 0, 1))):
        s.add_joint(nm, "Object", typ, axis=ax, armature=0.0)
    if kind == "reorient":
        s.add_geom("obj", "Object", "capsule", REORIENT_CAPS_100[0][:2])
        s.add_geom("top", "Object", "sphere", (0.002,), pos=(0, 0, -0.035))     # xml:36-37 (names as in the reference)
        s.add_geom("bot", "Object", "sphere", (0.002,), pos=(0, 0, 0.035))
    else:   # the pen: cylinder .015 x .065, density 1500; orientation markers are sites (myohand_pen.xml:34,40-41)
        mp = 1500.0 * math.pi * 0.015 ** 2 * 0.13
        ob = s.bodies[s._bname["Object"]]
        ob.mass = mp; ob.inertia = np.array([mp * (3 * 0.015 ** 2 + 0.13 ** 2) / 12.0] * 2 + [0.5 * mp * 0.015 ** 2])
        s.add_geom("obj", "Object", "cylinder", (0.015, 0.065))
        s.add_site("object_top", "Object", (0, 0, 0.065)); s.add_site("object_bottom", "Object", (0, 0, -0.065))
    s.add_site("eps_ball", "world", (OX, 0.004, OZ - 0.005))               # xml:24 vs :26: 5 mm below the object origin
    s.add_site("success", "world", (OX, -0.004, OZ + 0.2))
    s.add_body("target", "world", pos=(OX, -0.004, OZ + 0.2), quat=oq, mass=0.0)
    if kind == "reorient":
        s.add_geom("t_top", "target", "sphere", (0.002,), pos=(0, 0, -0.035))
        s.add_geom("t_bot", "target", "sphere", (0.002,), pos=(0, 0, 0.035))
    else:
        s.bodies[s._bname["target"]].quat = np.array([1.0, 0, 0, 0])            # myohand_pen.xml:44: no euler on the target
        s.add_site("target_top", "target", (0, 0, 0.065)); s.add_site("target_bottom", "target", (0, 0, -0.065))
    # contact dimension of the object's pairs = max of the two geoms (MuJoCo): the skin is condim 3 (xml:16), the pen is condim 4
    # (myohand_pen.xml:34: torsional friction, six pyramid rows per contact); the reorient object is authored condim 4 too
    # (myohand_sar.xml:36) but the env sets it back to 3 at every reset (reorient_sar_v0.py:425)
    for c in caps:
        s.add_contact_pair("obj", c, condim=3 if kind == "reorient" else 4, friction=(1.0, 0.005, 0.0001))
    if kind != "reorient":
        # six rows per contact.  60 rows still fit eight envs per CU (161 KB of LDS; 3.13 M env-steps/s at 2048 envs, as with 56); the
        # full 64-row table costs the eighth wave (2.01 M) -- tools/gpu_pen_njmax.py.  Under random actions about one env in 2048 has a
        # contact dropped (status bit 8) at either bound.
        s.njmax = 60
    return s




def make_hand_keyturn() -> ModelSpec:
    """myoHand + key (myosuite/envs/myo/assets/hand/myohand_keyturn.xml:22-32, "Index Thumb Model for turning key task"):
    the key is the LAST body, one hinge about its shaft (``keyjoint``: frictionloss 0.02, damping 0.1 -- xml:29) = the last
    qpos / qvel (key_turn_v0.py:86-91), geoms ellipsoid head .030 .030 .004, capsule shaft .005 x .070, box bit .015 .010 .004
    (xml:25-28), site ``keyhead`` at the body origin.  nq = nv = 24, 39 muscles, obs 93.  The key is placed between the open
    synthetic hand's index and thumb tips with the shaft pointing away from the wrist; index / thumb phalanx capsules collide
    with the three key geoms."""
    s = make_hand()
    s.name = "myohand_keyturn"
    s.nconmax = 6
    caps = _add_hand_capsules(s, only=("proxph2", "midph2", "distph2", "proximal_thumb", "distal_thumb"))
    KP = (0.388, -0.056, 0.970)
    s.add_body("key", "world", pos=KP, quat=(math.cos(math.pi / 2), 0.0, 0.0, math.sin(math.pi / 2)),   # key x = world -x
               mass=0.031, ipos=(-0.03, 0.001, 0), inertia=(4.0e-6, 6.0e-5, 6.2e-5))
    s.add_joint("keyjoint", "key", "hinge", axis=(1, 0, 0), damping=0.1, frictionloss=0.02)
    s.add_geom("keyhead", "key", "ellipsoid", (0.030, 0.030, 0.004))
    s.add_geom("keyshaft", "key", "capsule", (0.005, 0.070), pos=(-0.045, 0, 0), quat=(math.cos(1.57 / 2), 0.0, math.sin(1.57 / 2), 0.0))
    s.add_geom("keybit", "key", "box", (0.015, 0.010, 0.004), pos=(-0.1, 0.008, 0))
    s.add_site("keyhead", "key", (0, 0, 0))
    for c in caps:
        for kg in ("keyhead", "keyshaft", "keybit"):
            s.add_contact_pair(c, kg, condim=3, friction=(1.0, 0.005, 0.0001))
    return s


def make_hand_hold() -> ModelSpec:
    """myoHand + free-floating ellipsoid to hold (myosuite/envs/myo/assets/hand/myohand_hold.xml:14-23): object on a free
    joint (nq 30 / nv 29), frictionless contacts (condim 1), a world-fixed ``goal`` site and an ``object`` site.

    The object geom is authored ``condim="1"`` (xml:19).  A generated pair's contact dimension is the MAX over its two geoms, and the
    other one is a skin geom of the absent ``myo_sim`` tree: if those carry MuJoCo's default (3) the real contacts are frictional
    despite the attribute -- the importer applies the max rule to whatever the real files say; this stand-in follows the attribute."""
    s, caps = _hand_palm_up_with_capsules("myohand_hold")
    OX, OZ = 0.325, 1.0 + 0.009 + 0.032
    size = (0.025, 0.036, 0.030)
    m = 1000.0 * 4.0 / 3.0 * math.pi * size[0] * size[1] * size[2]
    s.add_body("object", "world", pos=(OX, 0.0, OZ), mass=m,
               inertia=(m / 5 * (size[1] ** 2 + size[2] ** 2), m / 5 * (size[0] ** 2 + size[2] ** 2), m / 5 * (size[0] ** 2 + size[1] ** 2)))
    s.add_joint("object_free", "object", "free")
    s.add_geom("object", "object", "ellipsoid", size)
    s.add_site("object", "object", (0, 0, 0))
    s.add_site("goal", "world", (OX - 0.005, -0.010, OZ + 0.020))          # xml:15 vs :18: goal - object = (-.005, -.01, +.02)
    for c in caps:
        s.add_contact_pair(c, "object", condim=1, friction=(1.0, 0.005, 0.0001))
    return s


def make_hand_reorient() -> ModelSpec:
    return _make_hand_with_object("reorient")


def make_hand_pen() -> ModelSpec:
    """myoHand + pen (myosuite/envs/myo/assets/hand/myohand_pen.xml:22-52): cylinder .015 x .065 on 3 slide + 3 hinge joints,
    orientation read through the object_top / object_bottom sites; collides with the hand capsules through the
    segment-vs-cylinder narrow phase."""
    return _make_hand_with_object("pen")


def _rest_pose_segments(s: ModelSpec, names):
    """World-frame axis segments (centre, unit axis, half length, radius) of the named capsules at qpos0 (every hand joint is a hinge
    with ref 0: the rest pose is the body frames composed down the tree)."""
    from .kin_np import quat_mul, quat2mat
    nb = len(s.bodies)
    xpos = np.zeros((nb, 3)); xquat = np.zeros((nb, 4)); xquat[0, 0] = 1.0
    for b in range(1, nb):
        bd = s.bodies[b]
        xpos[b] = xpos[bd.parent] + quat2mat(xquat[bd.parent][None])[0] @ bd.pos
        xquat[b] = quat_mul(xquat[bd.parent][None], np.asarray(bd.quat, float)[None])[0]
    out = {}
    for n in names:
        g = s.geoms[s._gname[n]]
        R = quat2mat(xquat[g["body"]][None])[0]
        c = xpos[g["body"]] + R @ g["pos"]
        u = (R @ quat2mat(np.asarray(g["quat"], float)[None])[0])[:, 2]
        out[n] = (c, u, float(g["size"][1]), float(g["size"][0]))
    return out


def _segment_distance(c1, u1, h1, c2, u2, h2):
    """distance between two segments c + t u, |t| <= h (clamped closest points)"""
    d = c1 - c2
    b = float(u1 @ u2); r1 = float(u1 @ d); r2 = float(u2 @ d)
    den = 1.0 - b * b
    t1 = np.clip((b * r2 - r1) / den, -h1, h1) if den > 1e-12 else 0.0
    t2 = np.clip(r2 + b * t1, -h2, h2)
    t1 = np.clip(b * t2 - r1, -h1, h1)
    return float(np.linalg.norm(c1 + t1 * u1 - c2 - t2 * u2))


def make_hand_dense() -> ModelSpec:
    """The reorient hand under MuJoCo's DEFAULT collision filter instead of a hand-picked pair list (round 6: the pair list may be
    longer than the wave).  `myohand_sar.xml:15-18` gives every geom of class ``reorient`` contype = conaffinity = 1, so
    `mj_collision` tests every skin geom against every other one unless the two sit on the same body or on parent and child; the
    real hand "self-collides" (docs/source/suite.rst:288).  Here: each of the 20 collision capsules (a capsule per metacarpal and
    phalanx + the carpal row) against every other one that passes that filter, minus the pairs that already interpenetrate in
    the rest pose (neighbouring segments across a joint: what a modeller's ``<exclude>`` removes, as `myo_sim`'s hand does for its
    skin), plus the object against all twenty.  Pairs are ordered by (body1, body2) like MuJoCo's broad phase emits them.
    Contact / row bounds are the reorient model's (nconmax 12, njmax 56: surplus contacts are dropped in pair order and
    flagged, status bit 8, by oracle and kernel alike)."""
    s = _make_hand_with_object("reorient")
    s.name = "myohand_dense"
    caps = [p["g2"] if p["g1"] == "obj" else p["g1"] for p in s.pairs]
    seg = _rest_pose_segments(s, caps)
    obj_pairs = {c: p for c, p in zip(caps, s.pairs)}
    body_of = {c: s.geoms[s._gname[c]]["body"] for c in caps}
    entries = []
    for i, c1 in enumerate(caps):
        for c2 in caps[i + 1:]:
            b1, b2 = body_of[c1], body_of[c2]
            if b1 == b2 or s.bodies[b1].parent == b2 or s.bodies[b2].parent == b1:
                continue                                                    # mj_collision's same-body / parent-child filter
            (x1, u1, h1, r1), (x2, u2, h2, r2) = seg[c1], seg[c2]
            if _segment_distance(x1, u1, h1, x2, u2, h2) < r1 + r2 + 5e-4:
                continue                                                    # <exclude>: touching in the rest pose
            entries.append((min(b1, b2), max(b1, b2), dict(g1=c1, g2=c2)))
    ob = s._bname["Object"]
    for c in caps:
        entries.append((min(body_of[c], ob), max(body_of[c], ob), None, c))
    s.pairs = []
    for e in sorted(entries, key=lambda e: (e[0], e[1])):
        if e[2] is None:
            s.pairs.append(obj_pairs[e[3]])
        else:
            s.add_contact_pair(e[2]["g1"], e[2]["g2"], condim=3, friction=(1.0, 0.005, 0.0001))
    return s


# ----------------------------------------------------------------------------- contact toy
def make_contact_toy() -> ModelSpec:
    """Small model that exercises every contact primitive of the engine (plane-sphere, plane-capsule,
    sphere-sphere, sphere-capsule, capsule-capsule), a free joint and a joint equality.  Test model only."""
    s = ModelSpec("contact_toy", timestep=0.002)
    s.add_geom("floor", "world", "plane", (0, 0, 0), quat=(math.cos(0.05), math.sin(0.05), 0.0, 0.0))   # 5.7 deg tilt
    s.add_body("log", "world", pos=(0, 0, 0.08), mass=1.5, inertia=(0.006, 0.006, 0.002))
    s.add_joint("log_free", "log", "free")
    s.add_geom("log_cap", "log", "capsule", (0.05, 0.12), quat=(math.cos(math.pi / 4), 0.0, math.sin(math.pi / 4), 0.0))
    s.add_body("ball", "world", pos=(0.03, 0.01, 0.30), mass=0.6, inertia=(0.0009, 0.0009, 0.0009))
    s.add_joint("ball_x", "ball", "slide", axis=(1, 0, 0), damping=0.1)
    s.add_joint("ball_y", "ball", "slide", axis=(0, 1, 0), damping=0.1)
    s.add_joint("ball_z", "ball", "slide", axis=(0, 0, 1), damping=0.1)
    s.add_geom("ball_s", "ball", "sphere", (0.06,))
    s.add_body("arm", "world", pos=(0.0, -0.25, 0.32), mass=0.8, ipos=(0, 0, -0.12), inertia=(0.004, 0.004, 0.0004))
    s.add_joint("arm_hinge", "arm", "hinge", axis=(1, 0, 0), range=(-1.2, 1.2), damping=0.05, armature=0.002)
    s.add_geom("arm_cap", "arm", "capsule", (0.03, 0.10), pos=(0, 0, -0.14))
    s.add_body("puck", "world", pos=(0.25, 0.01, 0.30), mass=0.4, inertia=(0.0004, 0.0004, 0.0004))
    s.add_joint("puck_x", "puck", "slide", axis=(1, 0, 0), range=(-0.4, 0.1), damping=0.2)
    s.add_geom("puck_s", "puck", "sphere", (0.05,))
    s.add_body("rider", "world", pos=(0.5, 0.3, 0.2), mass=0.2, inertia=(0.0002, 0.0002, 0.0002))
    s.add_joint("rider_z", "rider", "slide", axis=(0, 0, 1), damping=0.1, armature=0.01)
    s.add_equality_joint("rider_z", "arm_hinge", (0.0, 0.1, 0.05))
    s.add_contact_pair("floor", "log_cap", condim=3, friction=(0.8, 0.005, 0.0001))
    s.add_contact_pair("floor", "ball_s", condim=3, friction=(0.6, 0.005, 0.0001))
    s.add_contact_pair("ball_s", "log_cap", condim=3, friction=(0.7, 0.005, 0.0001))
    s.add_contact_pair("arm_cap", "log_cap", condim=3, friction=(0.9, 0.005, 0.0001))
    s.add_contact_pair("puck_s", "ball_s", condim=1, margin=0.002)
    return s


def make_tendon_limit_toy() -> ModelSpec:
    """myoElbow with length limits on two of its muscle tendons (one reaches its lower, one its upper bound inside the joint
    range) and a limited fixed tendon on the joint.  Test model for the tendon-limit constraint rows; not a reference asset."""
    s = make_elbow()
    s.name = "tendon_limit_toy"
    cm = make_elbow().compile()
    lr = cm.arrays["ACT_LENGTHRANGE"].reshape(-1, 2).astype(float)
    tid = {t.name: i for i, t in enumerate(s.tendons)}
    for name, lo_f, hi_f in (("TRIlong_tendon", 0.0, 0.7), ("BIClong_tendon", 0.35, 1.0)):
        i = cm.names["actuator"][name.replace("_tendon", "")]
        L0, L1 = lr[i]
        t = s.tendons[tid[name]]
        t.limited = True; t.range = (L0 + lo_f * (L1 - L0), L0 + hi_f * (L1 - L0)); t.margin = 0.001
    s.add_tendon("flex_stop", [("joint", "r_elbow_flex", 1.0)], limited=True, range=(0.3, 1.9), solref=(0.01, 1.0))
    return s


def make_tree_toy(kind: str = "star") -> ModelSpec:
    """Limit-rows-only test models of different dof-tree shapes (hinges with ranges, damping, armature, one motor per joint; no
    contacts, tendons or equalities), for the three routes of the engine's factorisation: ``star`` -- a 2-dof stem carrying three
    equal 2-dof branches (regular segment tree: the tree-sparse L'DL kernels), ``chain`` -- a 10-link pendulum (deeper than the
    sparse kernels' eight levels: general-row kernels, dense Cholesky), ``comb`` -- branches of unequal length (more elimination steps), ``free`` -- a free-floating trunk with two arms
    (eight levels, quaternion dofs).  Not reference assets."""
    s = ModelSpec(f"tree_toy_{kind}", timestep=0.002)
    n = [0]

    def link(parent, axis, length, pos):
        i = n[0]; n[0] += 1
        s.add_body(f"b{i}", parent, pos=pos, mass=0.3 + 0.05 * (i % 3), ipos=(0, 0, -0.5 * length),
                   inertia=(0.3 * length * length / 12 + 1e-4, 0.3 * length * length / 12 + 1e-4, 2e-4))
        s.add_joint(f"j{i}", f"b{i}", "hinge", axis=axis, range=(-0.9 + 0.1 * (i % 4), 0.8), damping=0.05 + 0.01 * i, armature=0.002)
        s.add_motor(f"m{i}", f"j{i}", gear=1.0 + 0.2 * (i % 5), ctrlrange=(-1.0, 1.0))
        return f"b{i}"

    ax = ((1, 0, 0), (0, 1, 0), (0.6, 0.8, 0))
    if kind == "free":
        # a free-floating trunk (6 dofs: depths 0..5) carrying two 2-link arms: eight tree levels, the deepest the sparse kernels take
        s.add_body("trunk", "world", pos=(0, 0, 1.0), mass=2.0, inertia=(0.02, 0.03, 0.025))
        s.add_joint("root", "trunk", "free")
        for k in range(2):
            q = link("trunk", ax[k], 0.15, (0.1 * (2 * k - 1), 0, -0.05))
            link(q, ax[k + 1], 0.15, (0, 0, -0.15))
        return s
    if kind == "chain":
        p = "world"
        for i in range(10):
            p = link(p, ax[i % 3], 0.12, (0, 0, 1.5) if i == 0 else (0, 0, -0.12))
        return s
    p = link("world", ax[0], 0.2, (0, 0, 1.0))
    p = link(p, ax[1], 0.2, (0, 0, -0.2))
    lengths = (2, 2, 2) if kind == "star" else (1, 2, 3)
    for k, m in enumerate(lengths):
        q = p
        for i in range(m):
            q = link(q, ax[(k + i) % 3], 0.15, (0.08 * (k - 1), 0, -0.2) if i == 0 else (0, 0, -0.15))
    return s


def make_friction_toy() -> ModelSpec:
    """Two-link arm with dry joint friction (``frictionloss``), limits, damping and torque motors, plus a slider coupled to the
    elbow by a joint equality.  Test model for the friction-loss constraint rows (Huber cost); not a reference asset."""
    s = ModelSpec("friction_toy", timestep=0.002)
    s.add_body("upper", "world", pos=(0, 0, 1.0), mass=1.2, ipos=(0.15, 0, 0), inertia=(0.002, 0.012, 0.012))
    s.add_joint("shoulder", "upper", "hinge", axis=(0, 1, 0), range=(-1.5, 1.5), damping=0.05, armature=0.005,
                frictionloss=0.6)
    s.add_body("fore", "upper", pos=(0.3, 0, 0), mass=0.8, ipos=(0.12, 0, 0), inertia=(0.001, 0.006, 0.006))
    s.add_joint("elbow", "fore", "hinge", axis=(0, 1, 0), range=(-0.2, 2.0), damping=0.02, armature=0.003,
                frictionloss=0.15, solreffriction=(0.01, 1.0), solimpfriction=(0.95, 0.99, 0.001, 0.5, 2.0))
    s.add_body("knob", "fore", pos=(0.25, 0, 0), mass=0.05, inertia=(2e-5, 2e-5, 2e-5))
    s.add_joint("knob_turn", "knob", "hinge", axis=(1, 0, 0), damping=0.001, armature=0.0005, frictionloss=0.02)
    s.add_body("slider", "world", pos=(0.5, 0.3, 0.5), mass=0.3, inertia=(3e-4, 3e-4, 3e-4))
    s.add_joint("slide_z", "slider", "slide", axis=(0, 0, 1), damping=0.2, armature=0.01)
    s.add_equality_joint("slide_z", "elbow", (0.0, 0.05))
    s.add_motor("m_shoulder", "shoulder", gear=4.0, ctrlrange=(-1.0, 1.0))
    s.add_motor("m_elbow", "elbow", gear=1.5, ctrlrange=(-1.0, 1.0))
    s.add_motor("m_knob", "knob_turn", gear=0.05, ctrlrange=(-1.0, 1.0))
    # a position servo and a velocity servo (affine bias: force = kp (ctrl - q) resp. kv (ctrl - qdot))
    s.add_general("p_elbow", joint="elbow", gainprm=(2.0,), biasprm=(0.0, -2.0, -0.05), ctrlrange=(-0.2, 2.0), forcerange=(-3.0, 3.0))
    s.add_general("v_slider", joint="slide_z", gainprm=(0.5,), biasprm=(0.0, 0.0, -0.5), ctrlrange=(-1.0, 1.0))
    # activation states without muscle dynamics: first-order filter and integrator
    s.add_general("f_shoulder", joint="shoulder", gainprm=(1.5,), ctrlrange=(-1.0, 1.0), dyntype="filter", dynprm=(0.03,))
    s.add_general("i_knob", joint="knob_turn", gainprm=(0.01,), ctrlrange=(-1.0, 1.0), dyntype="integrator")
    return s


def _ground_keyframes(cm):
    """Shift the root height of every keyframe so that the lowest foot sphere just touches the floor (z = 0)."""
    from . import kin_np as K
    km = K.KinModel(cm.arrays, cm.nq, cm.nv, cm.nbody)
    A = cm.arrays
    gt = A["GEOM_TYPE"]; gb = A["GEOM_BODYID"]; gp = A["GEOM_POS"].reshape(-1, 3).astype(np.float64)
    gs = A["GEOM_SIZE"].reshape(-1, 3).astype(np.float64)
    feet = sorted(set(int(g) for g in A["PAIR_GEOM2"]))
    xpos, xquat, _, _ = km.fk(cm.key_qpos)
    for k in range(cm.key_qpos.shape[0]):
        low = np.inf
        for g in feet:
            assert gt[g] == 2
            w, x, y, z = xquat[k, gb[g]]
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            low = min(low, (xpos[k, gb[g]] + R @ gp[g])[2] - gs[g, 0])
        cm.key_qpos[k, 2] -= low


def make_plane_toy() -> ModelSpec:
    """Test model for the multi-contact colliders (oracle/mmo_collision.inc): a box, a cylinder and an ellipsoid on free joints over
    a tilted plane (up to 4 / 4 / 1 contacts), a capsule on two slide joints that stays exactly parallel to a world-fixed
    capsule (two contacts), and a free capsule ("rod") over a world-fixed box ("anvil"): mjc_CapsuleBox's one or two contacts."""
    s = ModelSpec("plane_toy", timestep=0.002)
    s.add_geom("floor", "world", "plane", (0, 0, 0), quat=(math.cos(0.03), 0.0, math.sin(0.03), 0.0))    # 3.4 deg tilt about y
    s.add_body("box", "world", pos=(0.0, 0.0, 0.05), mass=0.8, inertia=(0.002, 0.003, 0.004))
    s.add_joint("box_free", "box", "free")
    s.add_geom("box_g", "box", "box", (0.08, 0.05, 0.03))
    s.add_body("cyl", "world", pos=(0.4, 0.0, 0.08), mass=0.6, inertia=(0.002, 0.002, 0.001))
    s.add_joint("cyl_free", "cyl", "free")
    s.add_geom("cyl_g", "cyl", "cylinder", (0.04, 0.06))
    s.add_body("ell", "world", pos=(-0.4, 0.0, 0.05), mass=0.5, inertia=(0.001, 0.002, 0.003))
    s.add_joint("ell_free", "ell", "free")
    s.add_geom("ell_g", "ell", "ellipsoid", (0.07, 0.05, 0.03))
    QY90 = (math.cos(math.pi / 4), 0.0, math.sin(math.pi / 4), 0.0)
    s.add_geom("rail", "world", "capsule", (0.02, 0.10), pos=(0.0, 0.5, 0.30), quat=QY90)
    s.add_body("bar", "world", pos=(0.0, 0.5, 0.34), mass=0.3, inertia=(0.0005, 0.0005, 0.0001))
    s.add_joint("bar_z", "bar", "slide", axis=(0, 0, 1), damping=0.2)
    s.add_joint("bar_x", "bar", "slide", axis=(1, 0, 0), damping=0.2)
    s.add_geom("bar_g", "bar", "capsule", (0.02, 0.06), quat=QY90)
    s.add_contact_pair("floor", "box_g", condim=3, friction=(0.8, 0.005, 0.0001))
    s.add_contact_pair("floor", "cyl_g", condim=3, friction=(0.6, 0.005, 0.0001))
    s.add_contact_pair("floor", "ell_g", condim=3, friction=(0.7, 0.005, 0.0001))
    s.add_contact_pair("rail", "bar_g", condim=3, friction=(0.9, 0.005, 0.0001))
    s.add_geom("anvil", "world", "box", (0.06, 0.04, 0.02), pos=(0.0, -0.5, 0.10))            # top face at z = 0.12
    s.add_body("rod", "world", pos=(0.0, -0.5, 0.135), mass=0.2, inertia=(0.0002, 0.0002, 0.00003), quat=QY90)
    s.add_joint("rod_free", "rod", "free")
    s.add_geom("rod_g", "rod", "capsule", (0.015, 0.05))
    s.add_contact_pair("rod_g", "anvil", condim=3, friction=(0.8, 0.005, 0.0001))
    return s


_CACHE = {}


def builders() -> dict:
    """name -> ModelSpec builder of every synthetic model (the one table get_model and the muscle-condition variants use)."""
    return {"elbow": make_elbow, "hand": make_hand, "leg": make_leg, "contact_toy": make_contact_toy,
            "hand_reorient": make_hand_reorient, "hand_pen": make_hand_pen,
            "hand_hold": make_hand_hold, "elbow_exo": make_elbow_exo, "finger": make_finger,
            "motorfinger": lambda: make_finger(motor=True), "torso": make_torso,
            "friction_toy": make_friction_toy, "hand_keyturn": make_hand_keyturn,
            "tendon_limit_toy": make_tendon_limit_toy, "hand_contact": lambda: make_hand(self_collision=True),
            "hand_dense": make_hand_dense,
            "leg_implicit": lambda: make_leg(implicit=True), "torso_exo": lambda: make_torso(exosuit=True),
            "tree_star": lambda: make_tree_toy("star"), "tree_chain": lambda: make_tree_toy("chain"),
            "tree_comb": lambda: make_tree_toy("comb"), "tree_free": lambda: make_tree_toy("free"), "plane_toy": make_plane_toy}


def compile_spec(name: str, edit=None) -> CompiledModel:
    """Build the named spec, optionally edit it (sarcopenia: base_v0.py:63-67), compile, attach keyframes."""
    spec = builders()[name]()
    if edit is not None:
        edit(spec)
    cm = spec.compile()
    keys = getattr(spec, "keys", None)
    if keys:   # keyframes (mjModel.key_qpos / key_qvel); host-side only, not part of the blob
        cm.key_qpos = np.array([k[0] for k in keys]); cm.key_qvel = np.array([k[1] for k in keys])
        if name in ("leg", "leg_implicit"):
            _ground_keyframes(cm)
    return cm


def get_model(name: str) -> CompiledModel:
    """Compiled synthetic model by short name ('elbow' | 'hand' | 'leg' | ...: see builders())."""
    if name not in _CACHE:
        _CACHE[name] = compile_spec(name)
    return _CACHE[name]
