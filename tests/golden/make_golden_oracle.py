"""Golden trajectories of OUR fp64 oracle on the synthetic models (regression pins; the reference
itself holds no mj_step golden vectors).  Run:  python tests/golden/make_golden_oracle.py
Inputs are fully determined by Philox streams: qpos0 ~ pose_reset_draws(env, episode 0, seed 0),
actions = uniform_stream(seed 0, stream_id = env-step index)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myosuite_amd.model import synth
from oracle import oracle as O
from oracle import env_oracle as EO

OUT = os.path.dirname(os.path.abspath(__file__))
NENV, NSTEPS, NSUB = 8, 30, 10


def rollout(name):
    cm = synth.get_model(name)
    om = O.OracleModel(cm)
    lo, hi = cm.jnt_range[:, 0], cm.jnt_range[:, 1]
    qpos = np.zeros((NSTEPS + 1, NENV, cm.nq)); qvel = np.zeros((NSTEPS + 1, NENV, cm.nv)); act = np.zeros((NSTEPS + 1, NENV, cm.na))
    ds = []
    for e in range(NENV):
        uq, _ = EO.pose_reset_draws(cm.nq, e, 0, 0)
        d = O.OracleData(om)
        d.qpos[:] = (lo + (hi - lo) * uq).astype(np.float32)
        ds.append(d)
        qpos[0, e] = d.qpos
    for s in range(NSTEPS):
        a = EO.uniform_stream(NENV * cm.nu, 0, s).reshape(NENV, cm.nu)
        ctrl = (1.0 / (1.0 + np.exp(-5.0 * (a.astype(np.float64) - 0.5)))).astype(np.float32)
        for e, d in enumerate(ds):
            d.ctrl[:] = ctrl[e]
            d.step(NSUB)
            qpos[s + 1, e] = d.qpos; qvel[s + 1, e] = d.qvel; act[s + 1, e] = d.act
    return dict(qpos=qpos, qvel=qvel, act=act, model_hash=np.array(cm.hash()))


def rollout_leg(nenv=6, nsteps=20):
    """myoLegWalk-v0 env-steps (10 substeps of 1 ms, foot contacts + knee equalities) from the stride keyframes with the
    walk-reset noise; actions 0.6 * uniform_stream(seed 0, stream_id = step)."""
    cm = synth.get_model("leg")
    orc = []
    qpos = np.zeros((nsteps + 1, nenv, cm.nq)); qvel = np.zeros((nsteps + 1, nenv, cm.nv)); act = np.zeros((nsteps + 1, nenv, cm.na))
    obs = np.zeros((nsteps + 1, nenv, 403)); dense = np.zeros((nsteps, nenv))
    for e in range(nenv):
        coin, z = EO.walk_reset_draws(cm.nq, e, 0, 0)
        k = 2 if coin < 0.5 else 3
        w = EO.WalkEnvOracle(cm)
        obs[0, e] = w.reset((cm.key_qpos[k].astype(np.float32) + z).astype(np.float64), cm.key_qvel[k])
        qpos[0, e] = w.d.qpos; qvel[0, e] = w.d.qvel
        orc.append(w)
    for s in range(nsteps):
        a = 0.6 * EO.uniform_stream(nenv * cm.nu, 0, s).reshape(nenv, cm.nu).astype(np.float64)
        for e, w in enumerate(orc):
            o, r, done, _ = w.step(a[e])
            obs[s + 1, e] = o; dense[s, e] = r
            qpos[s + 1, e] = w.d.qpos; qvel[s + 1, e] = w.d.qvel; act[s + 1, e] = w.d.act
    return dict(qpos=qpos, qvel=qvel, act=act, obs=obs, dense=dense, model_hash=np.array(cm.hash()))


def rollout_plane_toy(nenv=6, nsteps=160):
    """`plane_toy` (every primitive collider: plane vs box / cylinder / ellipsoid, parallel capsules, capsule vs box = the rod over
    the anvil): bodies dropped from random attitudes a few centimetres up (Generator seed 0), 160 substeps of falling, landing and
    settling; positions, velocities and the contact COUNT per substep (pins multiplicities: one or two capsule-box contacts, up to
    four plane-box / plane-cylinder contacts)."""
    cm = synth.get_model("plane_toy")
    om = O.OracleModel(cm)
    rng = np.random.default_rng(0)
    qpos = np.zeros((nsteps + 1, nenv, cm.nq)); qvel = np.zeros((nsteps + 1, nenv, cm.nv)); ncon = np.zeros((nsteps + 1, nenv), np.int32)
    ds = []
    for e in range(nenv):
        d = O.OracleData(om)
        q = cm.qpos0.astype(np.float64).copy()
        for k in range(3):                                   # box, cylinder, ellipsoid: lifted and tumbled
            o = 7 * k
            q[o + 2] += rng.uniform(0.01, 0.05)
            qq = rng.standard_normal(4) * (0.1 if e % 2 else 1.0) + np.array([1.0, 0, 0, 0]); q[o + 3:o + 7] = qq / np.linalg.norm(qq)
        q[21] = rng.uniform(0.0, 0.02); q[22] = rng.uniform(-0.1, 0.1)                       # the bar above the rail
        q[23] += rng.uniform(-0.04, 0.04); q[25] += rng.uniform(0.002, 0.03)                 # the rod above the anvil, slightly tilted / yawed
        th, yw = rng.uniform(-0.05, 0.05), rng.uniform(-0.4, 0.4)
        cy, sy, w1, z1 = np.cos(np.pi / 4 + th / 2), np.sin(np.pi / 4 + th / 2), np.cos(yw / 2), np.sin(yw / 2)
        q[26:30] = [w1 * cy, -z1 * sy, w1 * sy, z1 * cy]
        d.qpos[:] = q.astype(np.float32); d.qvel[:] = (0.2 * rng.standard_normal(cm.nv)).astype(np.float32)
        d.forward()
        ds.append(d); qpos[0, e] = d.qpos; qvel[0, e] = d.qvel; ncon[0, e] = d.ncon
    for s in range(nsteps):
        for e, d in enumerate(ds):
            d.step()
            qpos[s + 1, e] = d.qpos; qvel[s + 1, e] = d.qvel; ncon[s + 1, e] = d.ncon
    return dict(qpos=qpos, qvel=qvel, ncon=ncon, model_hash=np.array(cm.hash()))


if __name__ == "__main__":
    r = rollout_plane_toy()
    np.savez_compressed(os.path.join(OUT, "oracle_traj_plane_toy.npz"), **r)
    print("plane_toy", r["model_hash"], "max contacts", int(r["ncon"].max()), float(np.abs(r["qpos"][-1]).max()))
    r = rollout_leg()
    np.savez_compressed(os.path.join(OUT, "oracle_traj_leg.npz"), **r)
    print("leg", r["model_hash"], float(np.abs(r["qpos"][-1]).max()))
    for name in ("elbow", "hand"):
        r = rollout(name)
        np.savez_compressed(os.path.join(OUT, f"oracle_traj_{name}.npz"), **r)
        print(name, r["model_hash"], float(np.abs(r["qpos"][-1]).max()))
