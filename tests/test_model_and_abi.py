"""Host logic: blob packing, model compiler, and that the C-ABI library loads and exports every
symbol include/myosim.h declares (no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from myosuite_amd.model import blob, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_roundtrip(models):
    for cm in models.values():
        un = blob.unpack(cm.blob)
        for name, arr in cm.arrays.items():
            got = un[name].reshape(-1)
            ref = np.asarray(arr).reshape(-1)
            assert got.size == ref.size, name
            np.testing.assert_array_equal(got, ref.astype(got.dtype))
        assert cm.blob[0] == blob.MAGIC and cm.blob[3] == cm.blob.size


def test_dimensions_pinned_by_reference(models):
    # elbow 1/1/6, hand 23/23/39 (NPG policy pickles n=9,m=6 / n=108,m=39)
    e, h = models["elbow"], models["hand"]
    assert (e.nq, e.nv, e.nu, e.na) == (1, 1, 6, 6)
    assert (h.nq, h.nv, h.nu, h.na) == (23, 23, 39, 39)
    assert h.nbody == 30  # 29 bones + world (docs/source/suite.rst:88)
    assert list(h.names["joint"]) == synth.HAND_JOINTS
    assert list(h.names["actuator"]) == synth.HAND_MUSCLES
    assert abs(e.timestep - 0.002) < 1e-9 and abs(h.timestep - 0.002) < 1e-9


def test_compile_constants_sane(models):
    for cm in models.values():
        A = cm.arrays
        assert np.all(A["DOF_INVWEIGHT0"] > 0)
        lr = A["ACT_LENGTHRANGE"].reshape(-1, 2)
        assert np.all(lr[:, 1] > lr[:, 0])
        # tree ordering: parent < child and depth-first contiguity of dof subtrees
        par = A["BODY_PARENT"]
        assert np.all(par[1:] < np.arange(1, cm.nbody))
        dpar = A["DOF_PARENTID"]
        assert np.all(dpar < np.arange(cm.nv))


def test_hip_library_exports_header_symbols():
    from myosuite_amd import engine as E
    path = E.build()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, "include", "myosim.h")).read()
    names = sorted(set(re.findall(r"\b(mm_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/myosim.h but not exported"
    lib.mm_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.mm_version()


def test_ctypes_structs_match_header_field_order():
    from myosuite_amd import engine as E
    hdr = open(os.path.join(ROOT, "include", "myosim.h")).read()

    def c_fields(struct_end):
        body = hdr[:hdr.index(struct_end)]
        body = body[body.rindex("typedef struct {") + len("typedef struct {"):]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            for decl in stmt.split(","):
                out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", decl)[-1])
        return out

    for cls, end in ((E.mm_state, "} mm_state;"), (E.mm_derived, "} mm_derived;"), (E.mm_task, "} mm_task;"),
                     (E.mm_rollout, "} mm_rollout;")):
        assert [f[0] for f in cls._fields_] == c_fields(end), cls.__name__


def test_precision_enum_and_abi_version_match_header():
    from myosuite_amd import engine as E
    hdr = open(os.path.join(ROOT, "include", "myosim.h")).read()
    m = re.search(r"enum \{ MM_PREC_F32 = (\d+), MM_PREC_F64 = (\d+), MM_PREC_F64_STATE = (\d+) \}", hdr)
    assert m and tuple(int(x) for x in m.groups()) == (E.MM_PREC_F32, E.MM_PREC_F64, E.MM_PREC_F64_STATE)
    assert int(re.search(r"#define MM_ABI_VERSION (\d+)", hdr).group(1)) == E.MM_ABI_VERSION == E.lib().mm_abi_version()


def test_sarcopenia_model_compiles_for_every_registered_sarc_id():
    """registry registers a myoSarc* variant for every myo* id (myobase/__init__.py:25-31): each must have a model whose
    muscle peak forces are halved and nothing else changed (base_v0.py:63-67)."""
    from myosuite_amd.envs import registry, base_v0
    seen = set()
    for env_id, sp in registry.registry_specs().items():
        if sp["kwargs"].get("muscle_condition") != "sarcopenia":
            continue
        name = sp["kwargs"]["model"]
        if name in seen:
            continue
        seen.add(name)
        weak = base_v0._compiled_model(name, "sarcopenia"); base = base_v0._compiled_model(name, "")
        gw = weak.arrays["ACT_GAINPRM"].reshape(-1, 9); gb = base.arrays["ACT_GAINPRM"].reshape(-1, 9)
        np.testing.assert_allclose(gw[:, 2], 0.5 * gb[:, 2], err_msg=env_id)
        np.testing.assert_array_equal(np.delete(gw, 2, axis=1), np.delete(gb, 2, axis=1))
        assert float(np.abs(gb[:, 2]).max()) > 0
        np.testing.assert_array_equal(weak.arrays["ACT_BIASPRM"], base.arrays["ACT_BIASPRM"])
        for k in ("key_qpos", "key_qvel"):
            assert hasattr(weak, k) == hasattr(base, k)
    assert {"finger", "elbow_exo", "hand_keyturn", "torso", "hand", "leg", "elbow"} <= seen, seen


def test_no_cpu_fallback_without_gpu():
    import torch
    from myosuite_amd import engine as E
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(E.EngineError):
        E.HipModel(synth.get_model("elbow"))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "myosuite_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "liboracle" not in src, f
