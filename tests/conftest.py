import os
import sys
from pathlib import Path

import numpy as np
import pytest

# torch first: its wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever copy a
# process loads first serves both torch and libcloudtrace.so; torch on top of the system copy fails
# to find a GPU ("No HIP GPUs are available"), libcloudtrace.so works on either (bench.py loads torch
# first too).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing, the CPU tests do not need it
    torch = None

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = None


def pytest_collection_modifyitems(config, items):
    """GPU tests are only ever *selected* with -m gpu; if someone runs them on a box without a GPU
    they must fail loudly rather than pass on a fallback, so nothing is skipped here."""
    return


@pytest.fixture(scope="session")
def oracle_lib():
    import _oracle
    _oracle.build()
    return _oracle.lib()


@pytest.fixture(scope="session")
def product_lib():
    """libcloudtrace.so must exist (built by __graft_entry__.build / python -m deepestscatter_amd.build)."""
    from deepestscatter_amd import _lib, build
    if not _lib.LIB_PATH.exists():
        build.build()
    return _lib.load()


def sphere_volume(n=32, dims=None, radius=0.45, seed=None):
    """uint8 [Z,Y,X] texture (zero border) of a soft sphere, optionally non-cubic / with noise."""
    import _oracle
    nz, ny, nx = dims if dims else (n, n, n)
    z, y, x = np.mgrid[0:nz - 2, 0:ny - 2, 0:nx - 2].astype(np.float32)
    cz, cy, cx = (nz - 3) / 2, (ny - 3) / 2, (nx - 3) / 2
    r = np.sqrt(((x - cx) / (nx - 2)) ** 2 + ((y - cy) / (ny - 2)) ** 2 + ((z - cz) / (nz - 2)) ** 2)
    g = np.clip(1.0 - r / radius, 0.0, 1.0).astype(np.float32)
    if seed is not None:
        g *= np.random.default_rng(seed).random(g.shape, dtype=np.float32)
    return _oracle.quantize_volume(g)
