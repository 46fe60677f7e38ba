import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # outputs the kernels define completely (BFG_SHELL_OUT_OVERWRITE) start out as NaN in the tests, not as whatever the
    # allocator hands out: a pixel left unwritten fails every comparison
    os.environ.setdefault("BFG_POISON", "1")


def pytest_report_header(config):
    """what produced the golden fixtures: real healpy / pyccl or their stand-ins (tests/golden/make_golden.py --real-deps)"""
    import json
    import numpy as np
    lines = []
    for f in sorted(os.listdir(GOLDEN)):
        if f.endswith(".npz"):
            try:
                p = json.loads(str(np.load(os.path.join(GOLDEN, f))["_provenance"]))
                lines.append(f"golden {f}: deps={p.get('deps')} healpy={p.get('healpy')} | pyccl={p.get('pyccl')} | "
                             f"numba={p.get('numba')} | numpy {p.get('numpy')} scipy {p.get('scipy')}")
            except Exception:
                lines.append(f"golden {f}: no provenance record")
    return lines


def pytest_collection_modifyitems(config, items):
    """a plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them (the product has no
    CPU path to fall back to).  No torch at all counts as no GPU (the oracle / host tests do not need it)."""
    try:
        import torch
        n_gpu = torch.cuda.device_count()
    except ImportError:
        n_gpu = 0
    if n_gpu > 0:
        return
    skip = pytest.mark.skip(reason="no GPU visible: the HIP path has no CPU fallback")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


COSMO = {"Omega_m": 0.30, "Omega_b": 0.04, "h": 0.7, "sigma8": 0.8, "n_s": 0.96, "w0": -1.0}


@pytest.fixture(scope="session")
def cosmo():
    return dict(COSMO)
