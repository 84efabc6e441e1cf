"""Builds libcloudtrace.so (HIP, gfx950) in-tree with hipcc.  No cmake, no JIT cache.

    python -m deepestscatter_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels
to the GPU box with the working tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
ROOT = PKG.parent
LIB = PKG / "libcloudtrace.so"

SOURCES = ["ct_kernels.hip", "ct_group.hip", "ct_api.cpp", "ct_host.cpp"]
HEADERS = [CSRC / "ct_device.hpp", CSRC / "ct_internal.hpp", CSRC / "ct_exchange.hpp", ROOT / "include" / "cloudtrace.h",
           ROOT / "include" / "ct_fmath.h", PKG / "host" / "VdbReader.h"]

# -ffp-contract=off is part of the numeric contract (include/ct_fmath.h): results must be
# bit-identical to the CPU oracle, so no implicit FMA contraction and no fast-math.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-value"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC=...)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + HEADERS
    return any(d.stat().st_mtime > t for d in deps)


HOST = PKG / "host"
CLI = HOST / "cloudtrace"


def build_cli(force: bool = False, verbose: bool = False) -> Path:
    """The C++ host mirror of the reference's scene classes + headless CLI (g++, links the C ABI)."""
    srcs = [HOST / "main.cpp", HOST / "Cameras.h", HOST / "Scene.h", HOST / "SceneDescription.h", HOST / "VdbReader.h", HOST / "Collectors.h",
            ROOT / "include" / "cloudtrace.h"]
    if not force and CLI.exists() and LIB.exists() and all(s.stat().st_mtime <= CLI.stat().st_mtime for s in srcs + [LIB]):
        return CLI
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-pthread", "-o", str(CLI), str(HOST / "main.cpp"), f"-L{PKG}", "-lcloudtrace",
           "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return CLI


# Builds of the same sources beside the product's library (loaded with CT_LIBRARY=<file name>, deepestscatter_amd/_lib.py):
#   exp   -DCT_EXPERIMENTS: the product plus the measured-and-rejected experiments (the two path-exchange kernels of
#         csrc/ct_exchange.hpp; tests/test_exchange.py and the tools that A/B them)
#   w8    the DELTA kernel as two 1024-thread blocks per CU, 8 waves per SIMD, 64 VGPRs (round-4 A/B)
#   nofuse  both estimator kernels with a scatter phase and a march / tracking burst as separate scheduler iterations (round-4 A/B)
VARIANTS = {
    "exp": ["-DCT_EXPERIMENTS"],
    "w8": ["-DCT_DELTA_THREADS=1024", "-DCT_DELTA_WAVES=8"],
    "nofuse": ["-DCT_DELTA_FUSE=0", "-DCT_MARCH_FUSE=0", "-DCT_DELTA_CHECK_EVERY=1"],
    "noendmerge": ["-DCT_DELTA_END_MERGE=0"],
    "ab1": os.environ.get("CT_AB1_FLAGS", "").split(),   # scratch variants for A/B runs of compile-time switches
    "ab2": os.environ.get("CT_AB2_FLAGS", "").split(),
    "ab3": os.environ.get("CT_AB3_FLAGS", "").split(),
}


def build_variant(name: str, force: bool = False, verbose: bool = False) -> Path:
    out = PKG / f"libcloudtrace_{name}.so"
    deps = [CSRC / s for s in SOURCES] + HEADERS
    if force or not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        cmd = [hipcc(), *FLAGS, *VARIANTS[name], "-o", str(out), *[str(CSRC / s) for s in SOURCES], "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=str(CSRC))
    return out


def build(force: bool = False, verbose: bool = False) -> Path:
    if force or needs_build():
        # CT_EXTRA_FLAGS: e.g. -DCT_DEBUG_BOUNDS (device-side index checks that report instead of faulting)
        cmd = [hipcc(), *FLAGS, *os.environ.get("CT_EXTRA_FLAGS", "").split(), "-o", str(LIB), *[str(CSRC / s) for s in SOURCES], "-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=str(CSRC))
    build_cli(force=force, verbose=verbose)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:
        print(build_variant(sys.argv[sys.argv.index("--variant") + 1], force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
