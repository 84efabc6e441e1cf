"""AddressSanitizer + UndefinedBehaviorSanitizer runs of the HOST code (CPU only -- never on the GPU box):

* the CPU oracle (oracle/ct_oracle.c, `make -C oracle asan`): its own tests re-run in a child process on the sanitized build;
* the device-free part of the C ABI (csrc/ct_host.cpp + host/VdbReader.h: camera, quantiser, mipmaps, procedural cloud and
  the .vdb loader with its hand-written zip / Blosc / LZ4 decoders): a self-test against the reference loader's arithmetic
  and a hypothesis mutation fuzz of .vdb files (truncation, bit flips, length-field edits, spliced noise; the four
  compression modes) that must end in CT_OK or a CT_E_* code with a message, never in a sanitizer report;
* the header-only host classes that need no device (proto3 wire encoders / ScatterSample decoder of host/Collectors.h, the
  EXR writer of host/Exr.h): known bytes, round trips and a decoder fuzz.

The scheduler (csrc/ct_api.cpp) and the kernels need a device and are not covered: GPU sanitizers are not available on the
pool."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SAN = ROOT / "tests" / "sanitize"


def _gcc_file(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and Path(out).exists() else None


LIBASAN, LIBSTDCXX = _gcc_file("libasan.so"), _gcc_file("libstdc++.so")
pytestmark = pytest.mark.skipif(not (LIBASAN and LIBSTDCXX and shutil.which("make")), reason="gcc's libasan is not installed")


def _env(**extra):
    env = dict(os.environ)
    # libstdc++ beside libasan: the interpreter does not link it, and ASan's __cxa_throw interceptor needs the real one
    env["LD_PRELOAD"] = f"{LIBASAN} {LIBSTDCXX}"
    # no leak check (the interpreter's own allocations); an implausible allocation returns NULL -> std::bad_alloc -> CT_E_NOMEM
    env["ASAN_OPTIONS"] = "detect_leaks=0:allocator_may_return_null=1:max_allocation_size_mb=4096:abort_on_error=0"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env.update(extra)
    return env


def _clean(r):
    report = "AddressSanitizer" in r.stderr or "runtime error:" in r.stderr or "AddressSanitizer" in r.stdout
    assert r.returncode == 0 and not report, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-C", str(SAN), "-s"], check=True)
    return SAN / "_build"


def test_oracle_tests_pass_on_the_sanitized_build():
    subprocess.run(["make", "-C", str(ROOT / "oracle"), "-s", "asan"], check=True)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", str(ROOT / "tests" / "test_oracle_basics.py"),
                        str(ROOT / "tests" / "test_golden.py"), "-k", "not libm and not fixed_point"],
                       env=_env(CT_ORACLE_SANITIZE="1"), capture_output=True, text=True, cwd=str(ROOT), timeout=1500)
    _clean(r)
    assert " passed" in r.stdout and "libct_oracle_asan" not in r.stderr


def test_host_library_selftest_under_sanitizers(built):
    r = subprocess.run([sys.executable, str(SAN / "host_asan_driver.py"), "selftest"], env=_env(), capture_output=True, text=True, timeout=600)
    _clean(r)
    assert "selftest ok" in r.stdout


def test_vdb_loader_mutation_fuzz_under_sanitizers(built):
    r = subprocess.run([sys.executable, str(SAN / "host_asan_driver.py"), "fuzz", "1500"], env=_env(), capture_output=True, text=True, timeout=900)
    _clean(r)
    assert "rejected with a message" in r.stdout


def test_host_classes_under_sanitizers(built, tmp_path):
    exe = str(built / "host_classes_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([exe, "selftest", str(tmp_path)], env=env, capture_output=True, text=True, timeout=120)
    _clean(r)
    assert "selftest ok" in r.stdout
    r = subprocess.run([exe, "fuzz", "7", "50000"], env=env, capture_output=True, text=True, timeout=300)
    _clean(r)
    assert "decoded" in r.stdout
