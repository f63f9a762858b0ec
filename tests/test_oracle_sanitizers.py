"""The checker under AddressSanitizer + UndefinedBehaviorSanitizer ("-fsanitize=address,undefined build of the
oracle").  Every parity claim of this repo hangs on oracle/mmo_engine.c; this test steps every model family (limit rows, contacts of
every primitive pair kind, equalities, friction loss, tendon limits, all three integrators, per-env model deltas, the threaded batch
rollout) through `liboracle_asan.so` in a CHILD process with libasan preloaded and fails on any sanitizer report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import numpy as np, sys
sys.path.insert(0, %(root)r)
from oracle import oracle as O
from oracle import env_oracle as EO
from myosuite_amd.model import synth
assert O.variant() == "asan"
rng = np.random.default_rng(0)
names = ["elbow", "hand", "hand_contact", "hand_reorient", "hand_keyturn", "hand_pen", "hand_hold", "leg", "leg_implicit", "torso",
         "contact_toy", "plane_toy", "friction_toy", "tendon_limit_toy", "tree_free", "tree_comb"]
for name in names:
    cm = synth.get_model(name)
    om = O.OracleModel(cm)
    ds = [O.OracleData(om) for _ in range(3)]
    for e, d in enumerate(ds):
        if hasattr(cm, "key_qpos") and name.startswith("leg"):
            d.qpos[:] = cm.key_qpos[2]; d.qvel[:] = cm.key_qvel[2]
        else:
            lo, hi = cm.jnt_range[:, 0].astype(float), cm.jnt_range[:, 1].astype(float)
            fin = np.isfinite(lo) & np.isfinite(hi) & (hi > lo)
            q = np.array(d.qpos)
            nj = min(len(q), len(lo))
            q[:nj] = np.where(fin[:nj], lo[:nj] + (hi[:nj] - lo[:nj]) * rng.random(nj), q[:nj])
            if name not in ("tree_free", "hand_hold"):     # (free joints: keep the unit quaternion of qpos0)
                d.qpos[:] = q
        d.qvel[:] = 0.3 * rng.standard_normal(cm.nv)
        if name == "hand_reorient":
            gt, size, _, _ = EO.reorient_reset_draws(synth.reorient_tables("100"), e, 0, 0, 0.07)
            d.set_geom_size(cm.names["geom"]["obj"], size, gt)
    for d in ds:
        d.forward()
        for _ in range(12):
            d.ctrl[:] = rng.random(cm.nu)
            d.step()
        d.forward()
        assert np.all(np.isfinite(d.qpos)) and np.all(np.isfinite(d.qacc)), name
        d.full_M()
    acts = rng.random((4, len(ds), cm.nu))
    O.batch_rollout(om, ds, acts, nsub=3, nthreads=2, normalize=True, do_forward=True)
    print(name, "ok", ds[0].nefc, ds[0].ncon, flush=True)
# the narrow-phase test hook on every convex primitive
for gt, size in ((2, (0.03, 0, 0)), (3, (0.02, 0.05, 0)), (4, (0.02, 0.03, 0.05)), (5, (0.02, 0.04, 0)), (6, (0.02, 0.03, 0.04))):
    for _ in range(20):
        a = 0.08 * rng.standard_normal(3); u = rng.standard_normal(3); u /= np.linalg.norm(u)
        O.seg_shape(gt, size, a, u, 0.05)
print("SANITIZED-RUN-COMPLETE")
"""


def test_oracle_step_path_is_clean_under_asan_and_ubsan(tmp_path):
    from oracle import oracle as O
    O.build(variant="asan")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan.so next to this gcc")
    script = tmp_path / "asan_child.py"
    script.write_text(CHILD % {"root": ROOT})
    env = dict(os.environ, LD_PRELOAD=libasan, MYOSIM_ORACLE_VARIANT="asan",
               # CPython itself is not leak-clean; everything else is fatal
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    tail = (p.stdout[-2000:] + "\n" + p.stderr[-6000:])
    assert "AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr, tail
    assert p.returncode == 0 and "SANITIZED-RUN-COMPLETE" in p.stdout, tail
