"""The measured-and-rejected DELTA kernels that regroup paths by phase (csrc/ct_exchange.hpp) live in the EXPERIMENTS build of
the library (libcloudtrace_exp.so: python -m deepestscatter_amd.build --variant exp), not in the product's: their parity cases
(tests/experiments_cases.py) therefore run in a child process that loads that build (CT_LIBRARY), while this process keeps the
product's library.  The product's library must not contain them, and must say so when asked for them."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _kernel_names(path):
    out = subprocess.run(["strings", "-n", "12", str(path)], capture_output=True, text=True).stdout
    return {l.strip() for l in out.splitlines() if "_kernel" in l and l.startswith("_ZN2ct")}


def test_the_products_library_holds_the_products_kernels_only():
    from deepestscatter_amd import _lib, build
    names = _kernel_names(build.LIB)
    assert any("render_persistent_kernel" in n for n in names) and any("render_delta_kernel" in n for n in names)
    assert not any("render_delta_x_kernel" in n or "render_delta_w_kernel" in n for n in names)
    exp = build.PKG / "libcloudtrace_exp.so"
    if exp.exists():
        assert any("render_delta_x_kernel" in n for n in _kernel_names(exp))
    assert _lib.LIB_PATH.name == "libcloudtrace.so" or os.environ.get("CT_LIBRARY")


@pytest.mark.gpu
def test_exchange_kernels_in_the_experiments_build():
    from deepestscatter_amd import build
    build.build_variant("exp")
    env = dict(os.environ, CT_LIBRARY="libcloudtrace_exp.so")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", str(ROOT / "tests" / "experiments_cases.py")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_the_products_library_refuses_the_experiments_knob(monkeypatch):
    import deepestscatter_amd as ds
    from deepestscatter_amd import _lib
    monkeypatch.setenv("CT_EXCHANGE", "1")
    tex = np.zeros((16, 16, 16), np.uint8)
    tex[4:12, 4:12, 4:12] = 200
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=16, estimator=1)
    assert e.value.code == _lib.CT_E_INVAL and "experiments build" in e.value.message


@pytest.mark.gpu
def test_the_products_library_refuses_the_virtual_memory_bricks(monkeypatch):
    import deepestscatter_amd as ds
    from deepestscatter_amd import _lib
    tex = np.zeros((16, 16, 16), np.uint8)
    tex[4:12, 4:12, 4:12] = 200
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=16, flags=_lib.CT_FLAG_VMM_BRICKS)
    assert e.value.code == _lib.CT_E_INVAL and "experiments build" in e.value.message

