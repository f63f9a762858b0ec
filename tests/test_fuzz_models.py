"""Random small models through the HIP engine and the oracle: feature INTERACTIONS the hand / leg / toy models do not combine (random
trees of hinge and slide joints with springs, dampers, armature and reference angles; spatial tendons over several bodies with a
sphere or cylinder wrap, tendon springs with dead bands, tendon dampers and limits; muscles next to torque motors and position
servos; dry joint friction; a polynomial joint coupling).  Every model goes through the model compiler, the launch-width choice and
the kernel family its structure selects (tree-sparse or general rows)."""
import math

import numpy as np
import pytest

from myosuite_amd.model.spec import ModelSpec


def random_model(seed: int, integrator: int = 0, contacts: bool = False) -> ModelSpec:
    rng = np.random.default_rng(seed)
    s = ModelSpec(f"fuzz{seed}", timestep=0.002, integrator=integrator)
    lengths = []
    nlink = int(rng.integers(4, 10))
    names, parents = [], []
    for i in range(nlink):
        parent = "world" if i == 0 else names[int(rng.integers(max(0, i - 3), i))]          # bushy-ish tree, bounded depth growth
        length = float(rng.uniform(0.08, 0.2))
        pos = (0.0, 0.0, 1.2) if parent == "world" else tuple(rng.uniform(-0.03, 0.03, 2)) + (-float(rng.uniform(0.08, 0.2)),)
        s.add_body(f"b{i}", parent, pos=pos, mass=float(rng.uniform(0.2, 1.0)), ipos=(0.0, 0.0, -0.5 * length),
                   inertia=tuple(rng.uniform(3e-4, 3e-3, 3)))
        slide = rng.random() < 0.2
        ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
        kw = dict(axis=tuple(ax), damping=float(rng.uniform(0.01, 0.2)), armature=float(rng.uniform(5e-4, 5e-3)))
        if rng.random() < 0.5:
            kw.update(stiffness=float(rng.uniform(0.5, 5.0)), springref=float(rng.uniform(-0.3, 0.3)))
        if slide:
            s.add_joint(f"j{i}", f"b{i}", "slide", range=(-0.05, 0.06), **kw)
        else:
            s.add_joint(f"j{i}", f"b{i}", "hinge", range=(float(rng.uniform(-1.2, -0.3)), float(rng.uniform(0.3, 1.2))),
                        **({**kw, "frictionloss": 0.05} if (i == 2 and seed % 2 == 0) else kw))
        names.append(f"b{i}"); parents.append(parent); lengths.append(length)
        for k in range(2):
            s.add_site(f"s{i}_{k}", f"b{i}", tuple(rng.uniform(-0.03, 0.03, 2)) + (-float(rng.uniform(0.02, 0.9 * length)),))
    s.add_site("anchor", "world", (0.02, 0.01, 1.25))
    # tendons from the anchor (or a proximal body) down a chain of bodies; one of them wraps a sphere fixed to the first link
    s.add_geom("wrap_sph", "b0", "sphere", (0.025,), pos=(0.0, 0.0, -0.06))
    s.add_site("wrap_side", "b0", (0.05, 0.0, -0.06))
    ntend = int(rng.integers(2, 5))
    for t in range(ntend):
        leaf = int(rng.integers(1, nlink))
        chain = [leaf]
        while parents[chain[-1]] != "world":
            chain.append(names.index(parents[chain[-1]]))
        chain = chain[::-1][:4]                                     # root-side first, at most four bodies
        path = [("site", "anchor")] if t % 2 == 0 else []
        for ci, b in enumerate(chain):
            if t == 0 and ci == 1:
                path.append(("sphere", "wrap_sph", "wrap_side"))
            path.append(("site", f"s{b}_{t % 2}"))
        if len([p for p in path if p[0] == "site"]) < 2:
            path.append(("site", f"s{chain[-1]}_{(t + 1) % 2}"))
        kw = {}
        if t == 1:
            kw.update(stiffness=float(rng.uniform(20, 120)), damping=float(rng.uniform(0.2, 2.0)), springlength=(0.12, 0.2))
        if t == 2:
            kw.update(limited=True, range=(0.05, 0.45), margin=0.002)
        s.add_tendon(f"t{t}", path, **kw)
        if t != 2:
            s.add_muscle(f"mus{t}", f"t{t}", force=float(rng.uniform(20, 80)))
    for i in range(0, nlink, 3):
        s.add_motor(f"mot{i}", f"j{i}", gear=float(rng.uniform(0.3, 2.0)), ctrlrange=(-1.0, 1.0))
    if nlink >= 5:
        s.add_general("servo", joint="j4", gainprm=(1.5,), biasprm=(0.0, -1.5, -0.05), ctrlrange=(-0.5, 0.5))
    if seed % 3 == 0 and nlink >= 6:
        s.add_equality_joint("j5", "j1", (0.0, 0.3, 0.1))
    if contacts:
        # a collision capsule along every link, a floor the hanging limbs reach, and self-collision between links that are not neighbours
        s.add_geom("floor", "world", "plane", (0, 0, 0), pos=(0.0, 0.0, 0.88), quat=(math.cos(0.02), math.sin(0.02), 0.0, 0.0))
        for i in range(nlink):
            s.add_geom(f"cap{i}", f"b{i}", "capsule", (0.014, 0.4 * lengths[i]), pos=(0.0, 0.0, -0.5 * lengths[i]))
        for i in range(1, nlink):
            s.add_contact_pair("floor", f"cap{i}", condim=3, friction=(0.8, 0.005, 0.0001))
        npairs = 0
        for i in range(nlink):
            for j in range(i + 2, nlink):
                if parents[j] != names[i] and npairs < 6:
                    s.add_contact_pair(f"cap{i}", f"cap{j}", condim=3 if npairs % 2 else 1, friction=(0.6, 0.005, 0.0001))
                    npairs += 1
        s.nconmax = 12
    return s


@pytest.mark.parametrize("seed", range(10))
def test_random_models_compile_and_the_oracle_steps_them(oracle_lib, seed):
    """CPU half: the generator's models are valid (the oracle accepts the blob), stay finite under random excitation, and conserve
    nothing they should not (sanity: accelerations bounded).  The GPU half below compares the engines."""
    O = oracle_lib
    cm = random_model(seed).compile()
    d = O.OracleData(O.OracleModel(cm))
    rng = np.random.default_rng(seed)
    for _ in range(40):
        d.ctrl[:] = rng.uniform(-1, 1, cm.nu)
        d.step()
    assert np.all(np.isfinite(d.qpos)) and np.all(np.isfinite(d.qvel)) and np.abs(d.qacc).max() < 1e5 and d.warn == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(32))
def test_gpu_random_models_match_the_oracle(oracle_lib, seed):
    """one forward pass (poses, tendon lengths, actuator forces, constrained acceleration, row counts) and 20 free-running substeps
    of 24 random states per model, HIP engine vs oracle"""
    import torch
    from myosuite_amd import engine as E
    O = oracle_lib
    cm = random_model(seed).compile()
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(100 + seed)
    n = 24
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    q = (lo + (hi - lo) * (0.5 + 0.58 * (2 * rng.random((n, cm.nq)) - 1))).astype(np.float32)      # some joints past their limits
    v = rng.standard_normal((n, cm.nv)).astype(np.float32)
    act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.uniform(-0.5, 1.0, (n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
    if cm.na:
        st.act.copy_(torch.from_numpy(act))
    dv = E.Derived(hm, n, ["qacc", "nefc", "ten_length", "actuator_force", "xpos"])
    c = torch.from_numpy(ctrl).cuda().contiguous()
    E.forward(hm, st, c, dv)
    torch.cuda.synchronize()
    ga, gn = dv["qacc"].cpu().numpy().astype(np.float64), dv["nefc"].cpu().numpy()
    ds = []
    mism = 0
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.ctrl[:] = ctrl[e]
        if cm.na:
            d.act[:] = act[e]
        d.forward(); ds.append(d)
        np.testing.assert_allclose(dv["xpos"][e].cpu().numpy(), d.xpos, atol=2e-6)
        # (a wrap that grazes its sphere is ill-conditioned in fp32: the arc is r acos(c) with c -> 1; seed 22 has one at 3.9e-6 m)
        np.testing.assert_allclose(dv["ten_length"][e].cpu().numpy(), d.ten_length, atol=2e-5)
        np.testing.assert_allclose(dv["actuator_force"][e].cpu().numpy(), d.actuator_force, rtol=2e-4, atol=2e-4 * max(1.0, np.abs(d.actuator_force).max()))
        if d.nefc != gn[e]:
            mism += 1
            continue
        assert np.abs(ga[e] - d.qacc).max() < 1e-3 * max(1.0, np.abs(d.qacc).max()), (seed, e, np.abs(ga[e] - d.qacc).max(), np.abs(d.qacc).max())
    assert mism <= 1, mism
    for _ in range(2):
        E.step(hm, st, c, 10)
        for d in ds:
            d.step(10)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert int(st.status.cpu().max()) == 0 and max(d.warn for d in ds) == 0
    assert np.median(err) < 2e-5 and np.quantile(err, 0.9) < 5e-4, (seed, np.median(err), err.max())
    print(f"FUZZ seed {seed}: nv {cm.nv} ntendon {cm.ntendon} nu {cm.nu} neq {cm.neq} njmax {cm.njmax} kernel family {hm.info(E.INFO_KERNEL_FAMILY)} "
          f"lanes {hm.launch_lanes(n)} rows max {int(gn.max())} rollout err median {np.median(err):.1e} max {err.max():.1e}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,integrator", [(s_, i_) for s_ in (1, 4, 6, 13, 21) for i_ in (1, 3)], ids=lambda v: str(v))
def test_gpu_random_models_on_rk4_and_implicitfast(oracle_lib, seed, integrator):
    """the same generated models on the other two integrators (RK4 = 1, implicitfast = 3): kernels exist for the elbow- / hand- /
    leg-sized widths; a model whose width has no kernel of that integrator must be REFUSED (MM_EUNSUPPORTED), never stepped by
    another integrator"""
    import torch
    from myosuite_amd import engine as E
    O = oracle_lib
    cm = random_model(seed, integrator=integrator).compile()
    try:
        hm = E.HipModel(cm)
    except E.EngineError as exc:
        assert "unsupported" in str(exc).lower() or "kernel" in str(exc).lower() or "integrator" in str(exc).lower(), str(exc)
        pytest.skip(f"no kernel for nv {cm.nv} on integrator {integrator}: refused loudly ({exc})")
    om = O.OracleModel(cm)
    rng = np.random.default_rng(200 + seed)
    n = 16
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    q = (lo + (hi - lo) * (0.5 + 0.45 * (2 * rng.random((n, cm.nq)) - 1))).astype(np.float32)
    v = (0.5 * rng.standard_normal((n, cm.nv))).astype(np.float32)
    act = rng.random((n, cm.na)).astype(np.float32); ctrl = rng.uniform(-0.5, 1.0, (n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
    if cm.na:
        st.act.copy_(torch.from_numpy(act))
    c = torch.from_numpy(ctrl).cuda().contiguous()
    ds = []
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.ctrl[:] = ctrl[e]
        if cm.na:
            d.act[:] = act[e]
        ds.append(d)
    E.step(hm, st, c, 20)
    for d in ds:
        d.step(20)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert int(st.status.cpu().max()) == 0
    assert np.median(err) < 2e-5 and np.quantile(err, 0.9) < 5e-4, (seed, integrator, np.median(err), err.max())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 5, 12, 23])
def test_gpu_random_models_in_precision_mode_track_the_oracle_to_fp64_resolution(oracle_lib, seed):
    """MM_PREC_F64_STATE on generated models (tree-sparse, dense limit-row and general-row families): two independent fp64
    implementations of the pipeline agree to summation order over 20 free-running substeps -- or the model is refused loudly."""
    import torch
    from myosuite_amd import engine as E
    O = oracle_lib
    cm = random_model(seed).compile()
    try:
        hm = E.HipModel(cm, precision=E.MM_PREC_F64_STATE)
    except E.EngineError as exc:
        pytest.skip(f"no fp64 kernel for this structure: refused ({str(exc)[:80]})")
    om = O.OracleModel(cm)
    rng = np.random.default_rng(300 + seed)
    n = 8
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    q = lo + (hi - lo) * (0.5 + 0.5 * (2 * rng.random((n, cm.nq)) - 1))
    v = 0.5 * rng.standard_normal((n, cm.nv))
    act = rng.random((n, cm.na)); ctrl = rng.uniform(-0.5, 1.0, (n, cm.nu)).astype(np.float32)
    st = E.BatchState(hm, n)
    assert st.qpos.dtype == torch.float64
    st.qpos.copy_(torch.from_numpy(q)); st.qvel.copy_(torch.from_numpy(v))
    if cm.na:
        st.act.copy_(torch.from_numpy(act))
    ds = []
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q[e]; d.qvel[:] = v[e]; d.ctrl[:] = ctrl[e]
        if cm.na:
            d.act[:] = act[e]
        ds.append(d)
    E.step(hm, st, torch.from_numpy(ctrl).cuda().contiguous(), 20)
    for d in ds:
        d.step(20)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert err.max() < 1e-8, (seed, err)


def random_contact_scene(seed: int) -> ModelSpec:
    """three or four free bodies with one random primitive each (sphere / capsule / ellipsoid / cylinder / box) over a tilted plane,
    every body against the plane, spheres and capsules also against the other bodies (every pair kind the colliders implement)"""
    rng = np.random.default_rng(1000 + seed)
    s = ModelSpec(f"scene{seed}", timestep=0.002)
    tilt = float(rng.uniform(-0.05, 0.05))
    s.add_geom("floor", "world", "plane", (0, 0, 0), quat=(math.cos(tilt / 2), 0.0, math.sin(tilt / 2), 0.0))
    kinds = ["sphere", "capsule", "ellipsoid", "cylinder", "box"]
    nb = int(rng.integers(3, 5))
    chosen = [kinds[int(rng.integers(0, 5))] for _ in range(nb)]
    chosen[0] = "capsule"; chosen[1] = "sphere" if seed % 2 else "box"
    for i, k in enumerate(chosen):
        size = {"sphere": (0.03,), "capsule": (0.02, 0.05), "ellipsoid": (0.05, 0.035, 0.025), "cylinder": (0.03, 0.04), "box": (0.05, 0.035, 0.025)}[k]
        s.add_body(f"o{i}", "world", pos=(0.12 * i, 0.0, 0.06), mass=float(rng.uniform(0.2, 0.8)), inertia=tuple(rng.uniform(2e-4, 1e-3, 3)))
        s.add_joint(f"f{i}", f"o{i}", "free")
        s.add_geom(f"g{i}", f"o{i}", k, size)
        s.add_contact_pair("floor", f"g{i}", condim=3 if i % 2 == 0 else 1, friction=(float(rng.uniform(0.4, 1.0)), 0.005, 0.0001))
    order = {"sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6}
    for i in range(nb):
        for j in range(i + 1, nb):
            a, b = (i, j) if order[chosen[i]] <= order[chosen[j]] else (j, i)
            if chosen[a] in ("sphere", "capsule"):                      # sphere / capsule vs anything of equal or higher type
                s.add_contact_pair(f"g{a}", f"g{b}", condim=3, friction=(0.7, 0.005, 0.0001))
    s.nconmax = 14
    return s


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_contact_scenes_match_the_oracle(oracle_lib, seed):
    """every collider in random company: 48 random poses per scene (bodies hovering at, resting on, pressed into the plane and each
    other), row counts and constrained accelerations against the oracle, then 50 free-running substeps (median error; contact onsets
    are discontinuities, so the tail is only bounded loosely)"""
    import torch
    from myosuite_amd import engine as E
    O = oracle_lib
    cm = random_contact_scene(seed).compile()
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(2000 + seed)
    n = 64
    nb = cm.nq // 7
    # states a simulation actually reaches: the bodies are dropped in a loose pile (random attitudes and spins) and the ORACLE lets
    # them fall, land on the plane and on each other, roll and settle; snapshots after 30...250 substeps are the test states
    q = np.tile(cm.qpos0.astype(np.float64), (n, 1)); vv = np.zeros((n, cm.nv))
    dsim = O.OracleData(om)
    for e in range(n):
        q0 = cm.qpos0.astype(np.float64).copy()
        for i in range(nb):
            o = 7 * i
            q0[o:o + 3] = [rng.uniform(-0.03, 0.03) + 0.05 * (i % 2), rng.uniform(-0.03, 0.03), 0.07 + 0.075 * i + rng.uniform(0, 0.02)]
            qq = rng.standard_normal(4) * (0.05 if rng.random() < 0.3 else 1.0) + np.array([1.0, 0, 0, 0])
            q0[o + 3:o + 7] = qq / np.linalg.norm(qq)
        dsim.reset(); dsim.qpos[:] = q0; dsim.qvel[:] = 0.5 * rng.standard_normal(cm.nv)
        dsim.step(int(rng.integers(30, 250)))
        q[e] = dsim.qpos; vv[e] = dsim.qvel
    q32 = q.astype(np.float32); v = vv.astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q32)); st.qvel.copy_(torch.from_numpy(v))
    dv = E.Derived(hm, n, ["qacc", "nefc"])
    E.forward(hm, st, None, dv)
    torch.cuda.synchronize()
    ga, gn = dv["qacc"].cpu().numpy().astype(np.float64), dv["nefc"].cpu().numpy()
    ds, on, rel = [], np.zeros(n, int), np.zeros(n)
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q32[e]; d.qvel[:] = v[e]; d.forward(); ds.append(d)
        on[e] = d.nefc; rel[e] = np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max())
    flagged = np.array([d.warn != 0 for d in ds])                          # (contact / row bound hit: both sides drop, order-dependent)
    # Impacts at 1...2 m/s push a capsule or sphere centimetres into a soft contact; once its AXIS comes near the inside of a box or
    # cylinder the deepest point of the segment is a tie between two faces, and fp32 / fp64 may resolve it differently (the contact
    # normal then differs by 90 degrees) -- inherent to any closest-feature collider.  Envs with a sphere / capsule vs convex contact
    # deeper than a quarter radius only have to stay finite; plane contacts are exact at any depth.
    gt_, g1_, gs_ = cm.arrays["GEOM_TYPE"], cm.arrays["PAIR_GEOM1"], cm.arrays["GEOM_SIZE"].reshape(-1, 3)
    def _deep(d):
        return any(int(gt_[g1_[p_]]) != 0 and int(gt_[cm.arrays["PAIR_GEOM2"][p_]]) >= 4 and float(d.con_dist[c_]) < -0.25 * float(gs_[g1_[p_], 0])
                   for c_, p_ in enumerate(d.con_pair))
    deep = np.array([_deep(d) for d in ds])
    # A cylinder standing on its cap: mjc_PlaneCylinder places its rim contacts along the projection of the plane normal on the cap,
    # which for a cap parallel to the plane is a vector of length ~1e-6 -- its DIRECTION (where on the rim the three points sit) is
    # decided by rounding, in MuJoCo (threshold mjMINVAL = 1e-15) as here.  Same wrench up to the triangle's rotation; not compared.
    def _cap_down(d):
        return sum(1 for p_ in d.con_pair if int(gt_[g1_[p_]]) == 0 and int(gt_[cm.arrays["PAIR_GEOM2"][p_]]) == 5) >= 3
    deep |= np.array([_cap_down(d) for d in ds])
    mism = (on != gn) & ~flagged & ~deep
    ok = ~mism & ~flagged & ~deep
    assert np.all(np.isfinite(ga)) and ok.sum() >= n // 4 and ((on > 0) & ok).sum() >= 6, (int(ok.sum()), int(((on > 0) & ok).sum()))
    print(f"SCENE seed {seed}: nv {cm.nv} pairs {cm.npair} njmax {cm.njmax} lanes {hm.launch_lanes(n)} rows median {int(np.median(on))} max {on.max()} "
          f"mismatches {int(mism.sum())} flagged {int(flagged.sum())} deep {int(deep.sum())} compared {int(ok.sum())} ({int(((on > 0) & ok).sum())} with rows) rel |dqacc| median {np.median(rel[ok]):.1e} max {rel[ok].max():.1e}")
    gtn = {0: "plane", 2: "sphere", 3: "capsule", 4: "ellipsoid", 5: "cylinder", 6: "box"}
    for e in np.nonzero((rel > 3e-3) & ok)[0][:6]:
        d = ds[e]
        desc = [(gtn[int(cm.arrays["GEOM_TYPE"][cm.arrays["PAIR_GEOM1"][p_]])] + "-" + gtn[int(cm.arrays["GEOM_TYPE"][cm.arrays["PAIR_GEOM2"][p_]])],
                 round(float(d.con_dist[c_]), 5)) for c_, p_ in enumerate(d.con_pair)]
        print(f"   BAD env {e}: rel {rel[e]:.2e} rows {on[e]} niter {d.solver_niter} contacts {desc}")
    assert mism.sum() <= 2 and flagged.sum() <= n // 4, (int(mism.sum()), int(flagged.sum()))
    assert rel[ok].max() < 3e-3 and np.quantile(rel[ok], 0.9) < 3e-4, (rel[ok].max(), np.quantile(rel[ok], 0.9))
    E.step(hm, st, torch.zeros(n, 0, device="cuda"), 50)
    for d in ds:
        d.step(50)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert np.all(np.isfinite(err)) and np.median(err[ok]) < 1e-4, (seed, np.median(err[ok]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_random_articulated_models_with_contacts_match_the_oracle(oracle_lib, seed):
    """the generated trees again, now with a collision capsule per link, a tilted floor their limbs reach and self-collision between
    non-neighbouring links (the structure of the self-colliding hand / the legs, in random shapes): states reached by letting the
    ORACLE run under random excitations, then row counts and constrained accelerations of every state and 20 free-running substeps"""
    import torch
    from myosuite_amd import engine as E
    O = oracle_lib
    cm = random_model(seed, contacts=True).compile()
    hm = E.HipModel(cm); om = O.OracleModel(cm)
    rng = np.random.default_rng(500 + seed)
    n = 48
    q = np.zeros((n, cm.nq)); v = np.zeros((n, cm.nv)); act = np.zeros((n, cm.na)); ctrl = rng.uniform(-0.5, 1.0, (n, cm.nu)).astype(np.float32)
    lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
    dsim = O.OracleData(om)
    for e in range(n):
        dsim.reset(); dsim.qpos[:] = lo + (hi - lo) * rng.random(cm.nq); dsim.qvel[:] = rng.standard_normal(cm.nv)
        for _ in range(int(rng.integers(5, 60))):
            dsim.ctrl[:] = ctrl[e]; dsim.step()
        q[e] = dsim.qpos; v[e] = dsim.qvel
        if cm.na:
            act[e] = dsim.act
    q32, v32, a32 = q.astype(np.float32), v.astype(np.float32), act.astype(np.float32)
    st = E.BatchState(hm, n)
    st.qpos.copy_(torch.from_numpy(q32)); st.qvel.copy_(torch.from_numpy(v32))
    if cm.na:
        st.act.copy_(torch.from_numpy(a32))
    dv = E.Derived(hm, n, ["qacc", "nefc"])
    c = torch.from_numpy(ctrl).cuda().contiguous()
    E.forward(hm, st, c, dv)
    torch.cuda.synchronize()
    ga, gn = dv["qacc"].cpu().numpy().astype(np.float64), dv["nefc"].cpu().numpy()
    ds, on, rel, ncon = [], np.zeros(n, int), np.zeros(n), np.zeros(n, int)
    for e in range(n):
        d = O.OracleData(om); d.qpos[:] = q32[e]; d.qvel[:] = v32[e]; d.ctrl[:] = ctrl[e]
        if cm.na:
            d.act[:] = a32[e]
        d.forward(); ds.append(d)
        on[e], ncon[e] = d.nefc, d.ncon
        rel[e] = np.abs(ga[e] - d.qacc).max() / max(1.0, np.abs(d.qacc).max())
    flagged = np.array([d.warn != 0 for d in ds])
    mism = (on != gn) & ~flagged
    ok = ~mism & ~flagged
    print(f"ARTIC seed {seed}: nv {cm.nv} pairs {cm.npair} njmax {cm.njmax} lanes {hm.launch_lanes(n)} rows median {int(np.median(on))} max {on.max()} "
          f"states with contacts {int((ncon > 0).sum())} mismatches {int(mism.sum())} flagged {int(flagged.sum())} rel |dqacc| median {np.median(rel[ok]):.1e} max {rel[ok].max():.1e}")
    assert (ncon > 0).sum() >= n // 6 and mism.sum() <= 2 and flagged.sum() <= n // 4
    assert rel[ok].max() < 3e-3 and np.quantile(rel[ok], 0.9) < 3e-4, (rel[ok].max(), np.quantile(rel[ok], 0.9))
    E.step(hm, st, c, 20)
    for d in ds:
        d.ctrl[:] = d.ctrl; d.step(20)
    err = np.abs(st.qpos.cpu().numpy() - np.array([d.qpos for d in ds])).max(axis=1)
    assert np.all(np.isfinite(err)) and np.median(err[ok]) < 5e-5, (seed, np.median(err[ok]))


@pytest.mark.parametrize("seed,contacts,integrator", [(s_, s_ % 2 == 1, (0, 1, 3)[s_ % 3]) for s_ in range(12)])
def test_random_models_survive_the_mjcf_round_trip(oracle_lib, seed, contacts, integrator):
    """generated model -> MJCF text (mjcf.dump) -> importer (mjcf.load) -> compile: same dimensions and names, and the oracle steps both
    to bit-identical states -- springs, reference angles, tendon springs / limits / wraps, dry friction, couplings, servos, link
    capsules, contact pairs, nconmax, all three integrators through the XML.  The XML nests bodies, so a bushy tree comes back in
    depth-first order (MuJoCo's own numbering): joints / actuators are matched by NAME, and the permuted model's trajectory agrees to
    summation-order rounding (1e-9 after 25 steps), not to the bit; a tree that is already depth-first is bit-identical."""
    from myosuite_amd.model import mjcf
    O = oracle_lib
    spec = random_model(seed, integrator=integrator, contacts=contacts)
    cm0 = spec.compile()
    cm1 = mjcf.load(mjcf.dump(spec)).compile()
    if list(cm0.names["joint"]) != list(cm1.names["joint"]):
        # muscle length ranges left to the compiler are ESTIMATED from seeded joint-space samples whose columns follow the joint
        # order (spec.py compile(): lengthrange_samples), so a renumbered tree gets estimates that differ by ~1e-5; write the
        # compiled ranges into the XML (MuJoCo's `lengthrange` attribute) as a saved model would carry them
        lr = np.asarray(cm0.arrays["ACT_LENGTHRANGE"], np.float64)
        for i, a in enumerate(spec.actuators):
            if a.lengthrange is None and lr[i, 1] > lr[i, 0]:
                a.lengthrange = (float(lr[i, 0]), float(lr[i, 1]))
        cm0 = spec.compile()
        cm1 = mjcf.load(mjcf.dump(spec)).compile()
    for k in ("nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "ntendon", "neq", "npair", "njmax", "nconmax"):
        assert getattr(cm0, k) == getattr(cm1, k), k
    jn0, jn1, an0, an1 = list(cm0.names["joint"]), list(cm1.names["joint"]), list(cm0.names["actuator"]), list(cm1.names["actuator"])
    assert sorted(jn0) == sorted(jn1) and sorted(an0) == sorted(an1)
    assert cm0.nq == cm0.njnt and cm0.nv == cm0.njnt                      # hinge / slide joints only: qpos index = joint index
    pj = np.array([jn0.index(n) for n in jn1], int); pa = np.array([an0.index(n) for n in an1], int)
    same_order = bool((pj == np.arange(len(pj))).all() and (pa == np.arange(len(pa))).all())
    d0, d1 = O.OracleData(O.OracleModel(cm0)), O.OracleData(O.OracleModel(cm1))
    rng = np.random.default_rng(seed)
    lo, hi = cm0.jnt_range[:, 0].astype(float), cm0.jnt_range[:, 1].astype(float)
    q = lo + (hi - lo) * rng.random(cm0.nq); v = 0.3 * rng.standard_normal(cm0.nv); act = rng.random(cm0.na); ctrl = rng.uniform(-0.5, 1, cm0.nu)
    d0.qpos[:] = q; d0.qvel[:] = v; d0.ctrl[:] = ctrl
    d1.qpos[:] = q[pj]; d1.qvel[:] = v[pj]; d1.ctrl[:] = ctrl[pa]
    if cm0.na:
        ad0, ad1 = np.asarray(cm0.arrays["ACT_ACTADR"]), np.asarray(cm1.arrays["ACT_ACTADR"])     # activation slot of each actuator (-1: none)
        d0.act[:] = act
        for i1, i0 in enumerate(pa):
            assert (ad0[i0] < 0) == (ad1[i1] < 0)
            if ad1[i1] >= 0:
                d1.act[ad1[i1]] = act[ad0[i0]]
    d0.step(25); d1.step(25)
    err = np.abs(d0.qpos[pj] - d1.qpos).max()
    assert err <= (0.0 if same_order else 1e-9), err
    assert d0.nefc == d1.nefc and d0.ncon == d1.ncon
