"""Pin the engine against libmujoco where it is available (NOT runnable in the authoring container or on
the GPU boxes of this project: `mujoco` is not installed and there is no network).

    python tests/tools/validate_against_mujoco.py [--write-fixture] [--model <any name of model/synth.py builders(): hand|elbow|leg|plane_toy|hand_reorient|hand_contact|...> | --xml path.xml] [--steps 200] [--gpu]

1. writes the model as MJCF (`myosuite_amd.model.mjcf.dump`) or takes an MJCF file (e.g. the real myo_sim models) and imports it
   with `mjcf.load`;
2. loads the SAME XML with `mujoco.MjModel.from_xml_path`, copies the derived constants MuJoCo computes at compile time into the
   report (actuator_lengthrange / acc0, dof_invweight0, body_invweight0, tendon_invweight0, stat.meaninertia) next to ours;
3. steps both from identical (qpos, qvel, act, ctrl) with a fixed random ctrl sequence and reports per-stage differences after
   mj_forward (xpos, ten_length, actuator_force, qfrc_bias, qacc_smooth, efc count, qacc) and the state divergence over `--steps`
   mj_step calls: mujoco (fp64) vs oracle (fp64) vs, with --gpu, the HIP engine (fp32).
The report is what would turn "PARITY UNPINNED"  into pinned parity.
4. --write-fixture stores what libmujoco computed -- compile-time constants, every per-stage field of one mj_forward, a
   `--steps`-step (qpos, qvel, act) trajectory, the ctrl stream and (round 5) libmujoco's CONTACT LIST (count, distance, position,
   normal, in detection order) at the start, the middle and the end of the trajectory: `--model plane_toy` pins every primitive
   collider incl. mjc_CapsuleBox's one-or-two contacts -- as tests/golden/mujoco_<model>.npz.  Committed, that file
   makes tests/test_golden.py::test_oracle_matches_libmujoco_fixture (CPU) and
   tests/test_gpu_parity.py::test_hip_matches_libmujoco_fixture (GPU) run against libmujoco's numbers on hosts WITHOUT mujoco:
   one run by anyone who has `pip install mujoco` flips the engine rows of the scope table from "unpinned" to pinned.
"""
import argparse
import json
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


# mjData fields of one mj_forward stored in the fixture (same names in the oracle's OracleData)
FORWARD_FIELDS = ["xpos", "xquat", "xipos", "site_xpos", "subtree_com", "cdof", "cvel", "ten_length", "ten_velocity", "actuator_length",
                  "actuator_velocity", "actuator_force", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qfrc_smooth", "qacc_smooth",
                  "qfrc_constraint", "qacc"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="hand")
    ap.add_argument("--xml", default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--write-fixture", action="store_true")
    args = ap.parse_args()
    try:
        import mujoco
    except ImportError:
        print("mujoco is not installed: nothing to validate against (see module docstring)")
        return 2
    from myosuite_amd.model import mjcf, synth
    from oracle import oracle as O
    if args.xml:
        path = args.xml
        spec = mjcf.load(path)
    else:
        spec = synth.builders()[args.model]()        # every synthetic model: incl. plane_toy (all primitive colliders, mjc_CapsuleBox), hand_contact, leg_implicit
        path = os.path.join(tempfile.mkdtemp(), f"{args.model}.xml")
        open(path, "w").write(mjcf.dump(spec))
        # everything below uses the model AS IMPORTED FROM THE XML libmujoco reads: MuJoCo numbers sites / geoms / tendons in
        # document order, the builder in insertion order (found by the fake-mujoco round trip: site_xpos rows did not line up)
        spec = mjcf.load(path)
    cm = spec.compile()
    mjm = mujoco.MjModel.from_xml_path(path); mjd = mujoco.MjData(mjm)
    rep = {"xml": path, "dims": {"nq": [cm.nq, mjm.nq], "nv": [cm.nv, mjm.nv], "nu": [cm.nu, mjm.nu], "ntendon": [cm.ntendon, mjm.ntendon]}}
    A = cm.arrays
    rep["compile_constants_maxabs_diff"] = {
        "dof_invweight0": float(np.abs(A["DOF_INVWEIGHT0"] - mjm.dof_invweight0).max()),
        "body_invweight0": float(np.abs(A["BODY_INVWEIGHT0"].reshape(-1, 2) - mjm.body_invweight0).max()),
        "tendon_invweight0": float(np.abs(A["TENDON_INVWEIGHT0"] - mjm.tendon_invweight0).max()) if cm.ntendon else 0.0,
        "actuator_acc0": float(np.abs(A["ACT_ACC0"] - mjm.actuator_acc0).max()) if cm.nu else 0.0,
        "actuator_lengthrange": float(np.abs(A["ACT_LENGTHRANGE"].reshape(-1, 2) - mjm.actuator_lengthrange).max()) if cm.nu else 0.0,
        "meaninertia": float(abs(A["OPT_F"][2] - mjm.stat.meaninertia))}
    om = O.OracleModel(cm); d = O.OracleData(om)
    rng = np.random.default_rng(0)
    q0 = np.asarray(spec.keys[2][0], float) if hasattr(spec, "keys") and len(spec.keys) > 2 else cm.qpos0.astype(np.float64)
    v0 = rng.standard_normal(cm.nv) * 0.2
    a0 = rng.random(cm.na)
    ctrl = rng.random((args.steps, cm.nu))
    mjd.qpos[:] = q0; mjd.qvel[:] = v0; mjd.act[:] = a0; mjd.ctrl[:] = ctrl[0]
    d.qpos[:] = q0; d.qvel[:] = v0; d.act[:] = a0; d.ctrl[:] = ctrl[0]
    mujoco.mj_forward(mjm, mjd); d.forward()
    fix = None
    if args.write_fixture:
        # the fixture carries the MJCF text itself: its readers import exactly the model libmujoco compiled (include-free XML
        # only; for --xml files with <include> the readers need the same tree, so the text is stored for single-file models)
        xml_text = open(path).read()
        fix = dict(model=np.array(args.model if not args.xml else os.path.basename(args.xml)), model_hash=np.array(cm.hash()),
                   xml=np.array(xml_text if "<include" not in xml_text else ""),
                   mujoco_version=np.array(mujoco.__version__), q0=q0, v0=v0, a0=a0, ctrl=ctrl,
                   c_dof_invweight0=np.array(mjm.dof_invweight0), c_body_invweight0=np.array(mjm.body_invweight0),
                   c_tendon_invweight0=np.array(mjm.tendon_invweight0), c_actuator_acc0=np.array(mjm.actuator_acc0),
                   c_actuator_lengthrange=np.array(mjm.actuator_lengthrange), c_meaninertia=np.array(mjm.stat.meaninertia))
        for f in FORWARD_FIELDS:
            fix["f_" + f] = np.array(getattr(mjd, f)).copy()
        fix["f_nefc"] = np.array(int(mjd.nefc))

    def contacts(tag):
        """the contact list libmujoco holds right now (detection order): pins the colliders -- multiplicity, distance, position, normal"""
        n = int(mjd.ncon)
        fix[f"k{tag}_ncon"] = np.array(n)
        fix[f"k{tag}_dist"] = np.array([mjd.contact[i].dist for i in range(n)], float)
        fix[f"k{tag}_pos"] = np.array([np.array(mjd.contact[i].pos) for i in range(n)], float).reshape(n, 3)
        fix[f"k{tag}_normal"] = np.array([np.array(mjd.contact[i].frame)[:3] for i in range(n)], float).reshape(n, 3)
        fix[f"k{tag}_nefc"] = np.array(int(mjd.nefc))
    if fix is not None:
        contacts(0)
    fw = {}
    for ours, theirs in (("xpos", "xpos"), ("ten_length", "ten_length"), ("actuator_force", "actuator_force"),
                         ("qfrc_bias", "qfrc_bias"), ("qacc_smooth", "qacc_smooth"), ("qacc", "qacc")):
        x, y = np.asarray(getattr(d, ours)).ravel(), np.asarray(getattr(mjd, theirs)).ravel()
        fw[ours] = float(np.abs(x - y).max() / max(1e-12, np.abs(y).max())) if x.size else 0.0
    fw["nefc"] = [int(d.nefc), int(mjd.nefc)]
    rep["forward_rel_diff"] = fw
    hip = None
    if args.gpu:
        import torch
        from myosuite_amd import engine as E
        hm = E.HipModel(cm); st = E.BatchState(hm, 1)
        st.qpos.copy_(torch.from_numpy(q0.astype(np.float32))[None]); st.qvel.copy_(torch.from_numpy(v0.astype(np.float32))[None])
        st.act.copy_(torch.from_numpy(a0.astype(np.float32))[None])
        hip = (hm, st, E, torch)
    div = {"oracle_vs_mujoco": [], "hip_vs_mujoco": []}
    traj = {"qpos": [], "qvel": [], "act": []}
    for s in range(args.steps):
        mjd.ctrl[:] = ctrl[s]; d.ctrl[:] = ctrl[s]
        mujoco.mj_step(mjm, mjd); d.step()
        traj["qpos"].append(np.array(mjd.qpos)); traj["qvel"].append(np.array(mjd.qvel)); traj["act"].append(np.array(mjd.act))
        if fix is not None and s + 1 in (args.steps // 2, args.steps):
            # (mj_step leaves the contacts of the state it STARTED from: the readers compare after the same call)
            contacts(s + 1)
        div["oracle_vs_mujoco"].append(float(np.abs(d.qpos - mjd.qpos).max()))
        if hip:
            hm, st, E, torch = hip
            E.step(hm, st, torch.from_numpy(ctrl[s].astype(np.float32))[None].cuda().contiguous(), 1)
            div["hip_vs_mujoco"].append(float(np.abs(st.qpos[0].cpu().numpy() - mjd.qpos).max()))
    rep["qpos_divergence"] = {k: {"after_10": v[9] if len(v) > 9 else None, "final": v[-1], "max": max(v)} for k, v in div.items() if v}
    print(json.dumps(rep, indent=1))
    if fix is not None:
        for k, v in traj.items():
            fix["t_" + k] = np.array(v)
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", f"mujoco_{str(fix['model']).replace('.xml', '')}.npz")
        np.savez_compressed(out, **fix)
        print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
