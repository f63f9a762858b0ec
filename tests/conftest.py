import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests are the parity tests proper and need the device: skip (not fail) them on a host without one."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a HIP device (run on the GPU box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def models():
    from myosuite_amd.model import synth
    return {"elbow": synth.get_model("elbow"), "hand": synth.get_model("hand")}
