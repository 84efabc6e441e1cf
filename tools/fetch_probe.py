#!/usr/bin/env python3
"""Runs the PMC calibration probe (ct_debug_fetch_probe): `repeats` launches, each touching 2^k
distinct 128-byte lines once with the estimator's two unaligned 8-byte loads.  Use under
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/fetch_probe.py 25 3
and divide the kernel's FETCH_SIZE (KiB) by 2^k lines."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from deepestscatter_amd import _lib

k = int(sys.argv[1]) if len(sys.argv) > 1 else 25
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s = C.c_uint64(0)
_lib.check(_lib.load().ct_debug_fetch_probe(0, k, reps, C.byref(s)))
print(f"probe ok: 2^{k} lines x {reps} launches, checksum {s.value}")
