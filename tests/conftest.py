import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"     # exists only in the build container, never on the GPU box


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def modeldirs(tmp_path_factory):
    """Seeded synthetic model directories in the reference's on-disk format (see tools/gen_models.py)."""
    from tools import gen_models
    out = {}
    for fam in ("rife-v4.6", "rife-v2.3", "rife-v4", "rife-v3.1", "rife", "rife-HD"):
        out[fam] = gen_models.ensure(None, fam)
    return out


@pytest.fixture(scope="session")
def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
