"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol
include/cloudtrace.h declares, its host helpers agree with the oracle, and it refuses to compute
without a device (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import _oracle as O
import deepestscatter_amd as ds
from deepestscatter_amd import _lib
from conftest import sphere_volume

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol(product_lib):
    header = (ROOT / "include" / "cloudtrace.h").read_text()
    declared = re.findall(r"^CT_API [^(]*?\**(ct_[a-z_]+)\(", header, flags=re.M)
    assert len(declared) >= 24
    assert sorted(declared) == sorted(_lib.EXPORTS)
    for name in declared:
        assert hasattr(product_lib, name), name


def test_no_torch_or_cpp_types_in_the_abi():
    header = (ROOT / "include" / "cloudtrace.h").read_text()
    body = header.split("#ifdef __cplusplus")[1]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)       # comments may cite the reference's C++
    for banned in ("std::", "at::", "torch", "hipStream_t", "template"):
        assert banned not in body


def test_struct_layout_matches_header(product_lib):
    # sizeof(CtScene) as the C compiler sees it == ctypes layout (compile a probe with gcc)
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "cloudtrace.h"\nint main(){printf("%zu %zu", sizeof(CtScene), sizeof(CtCounters));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "p.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/p", f"{d}/p.c"], check=True)
        out = subprocess.run([f"{d}/p"], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == C.sizeof(_lib.CtScene)
    assert int(out[1]) == C.sizeof(_lib.CtCounters)


def test_camera_helper_matches_oracle():
    for aspect in (1.0, 2.0, 512 / 256, 37 / 21):
        for eye in ((2.5, -0.4, 0.0), (0.3, 1.9, -2.2)):
            got = ds.calculate_camera_variables(eye, (0, 0, 0), (0, 1, 0), 30.0, aspect)
            ref = O.camera_variables(eye, aspect=aspect)
            for g, r in zip(got, ref):
                assert np.array_equal(g, r)


def test_quantizer_and_mipmaps_match_oracle():
    rng = np.random.default_rng(11)
    for shape in ((7, 9, 11), (16, 16, 16), (1, 5, 3)):
        g = rng.random(shape, dtype=np.float32) ** 3
        t = ds.quantize_volume(g)
        assert np.array_equal(t, O.quantize_volume(g))
        a, b = ds.generate_mipmaps(t), O.generate_mipmaps(t)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_tile_owner_partitions_the_frame():
    for count in (1, 2, 3, 4, 8):
        masks = [ds.shard_mask(100, 60, i, count) for i in range(count)]
        total = np.sum(masks, axis=0)
        assert np.all(total == 1)
        sizes = [m.sum() for m in masks]
        assert max(sizes) - min(sizes) <= 0.15 * (100 * 60 / count) + 64
        for tx in range(13):
            for ty in range(8):
                o = ds.tile_owner(tx, ty, count)
                assert masks[o][ty * 8 if ty * 8 < 60 else 59, tx * 8 if tx * 8 < 100 else 99]


def test_procedural_cloud_is_deterministic_and_bordered():
    a = ds.make_procedural_cloud(48)
    b = ds.make_procedural_cloud(48)
    assert np.array_equal(a, b)
    assert a.shape == (48, 48, 48) and a.max() == 255
    assert a[0].max() == 0 and a[-1].max() == 0 and a[:, 0].max() == 0 and a[:, :, -1].max() == 0
    assert 0.02 < (a > 0).mean() < 0.5
    assert not np.array_equal(a, ds.make_procedural_cloud(48, seed=7))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_invalid_arguments_are_rejected_before_touching_the_gpu(product_lib):
    tex = sphere_volume(16)
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=16, mode=7)           # CloudMaterial.cpp:62
    assert e.value.code == _lib.CT_E_INVAL and "Invalid Render Mode" in e.value.message
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=16, shard_index=2, shard_count=2)
    assert e.value.code == _lib.CT_E_INVAL
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=5000)                  # seed packing x*4096+y
    assert e.value.code == _lib.CT_E_INVAL
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(tex, width=16, height=16, light_direction=(0, 0, 0))
    assert e.value.code == _lib.CT_E_INVAL
    assert product_lib.ct_destroy(None) == 0


def test_no_cpu_fallback_without_a_device(product_lib):
    if _has_gpu():
        pytest.skip("a GPU is present; the no-device path is exercised on the CPU container")
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.CloudTracer(sphere_volume(16), width=16, height=16)
    assert e.value.code == _lib.CT_E_NODEVICE
    assert "no CPU fallback" in e.value.message


def test_group_api_without_a_device_and_with_bad_arguments(product_lib):
    """ct_group_*: argument errors come back as CtStatus codes with a message, and without a GPU the group fails like a
    single handle does (no CPU fallback, no exception across the ABI)."""
    with pytest.raises(_lib.CloudTraceError) as e:
        ds.TracerGroup(sphere_volume(16), [], width=16, height=16)
    assert e.value.code == _lib.CT_E_INVAL
    if not _has_gpu():
        with pytest.raises(_lib.CloudTraceError) as e:
            ds.TracerGroup(sphere_volume(16), [0, 1], width=16, height=16)
        assert e.value.code == _lib.CT_E_NODEVICE and "shard 0" in e.value.message and "no CPU fallback" in e.value.message
    assert product_lib.ct_group_destroy(None) == 0
    n = C.c_uint32(0)
    assert product_lib.ct_group_size(None, C.byref(n)) == _lib.CT_E_INVAL


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under deepestscatter_amd/ may include, import,
    link or dlopen anything from oracle/."""
    pkg = ROOT / "deepestscatter_amd"
    for f in list(pkg.rglob("*")) + [ROOT / "include" / "cloudtrace.h", ROOT / "include" / "ct_fmath.h"]:
        if f.suffix in (".py", ".cpp", ".hip", ".hpp", ".h", ".c"):
            for line in f.read_text(errors="replace").splitlines():
                code = line.split("//")[0]
                if re.search(r"#\s*include.*oracle", code) or re.search(r"\b(import|from)\s+\S*oracle", code) \
                        or "libct_oracle" in code or "ct_oracle" in code.replace("oracle/ct_oracle.c", ""):
                    raise AssertionError(f"{f}: {line.strip()}")
    so = pkg / "libcloudtrace.so"
    if so.exists():
        assert b"ct_oracle" not in so.read_bytes()
