#!/usr/bin/env python3
"""Extract the two Lorenz-Mie phase tables (numeric data only) from the reference.

The reference keeps the tabulated phase function of a cloud droplet distribution
as two 4096-entry float literals lists (DeepestScatter_DataGen/.../src/Mie.cpp:8-4105
`mie`, :4107-8203 `choppedMie`; index i <-> cos(theta) = -1 + 2(i+0.5)/4096).
Those numbers are physical input data, not code.  This script parses them as
text (the reference is never compiled or imported) and writes them as one raw
little-endian float32 file:

    deepestscatter_amd/data/mie_raw.f32   = mie[4096] ++ choppedMie[4096]

Run once at development time (needs /root/reference); the output is committed.
"""
import re
import sys
from pathlib import Path

import numpy as np

REF = Path("/root/reference/DeepestScatter_DataGen/DeepestScatter_DataGen/src/Mie.cpp")
OUT = Path(__file__).resolve().parents[1] / "deepestscatter_amd" / "data" / "mie_raw.f32"


def main() -> int:
    text = REF.read_text()
    a = text.index("float_t mie[]")
    b = text.index("float_t choppedMie[]")
    c = text.index("getPhaseSampler")
    lit = re.compile(r"([-+]?\d+\.?\d*(?:[eE][-+]?\d+)?)f")
    mie = np.array([float(x) for x in lit.findall(text[a:b])], dtype=np.float32)
    chopped = np.array([float(x) for x in lit.findall(text[b:c])], dtype=np.float32)
    assert mie.shape == (4096,) and chopped.shape == (4096,), (mie.shape, chopped.shape)
    OUT.parent.mkdir(parents=True, exist_ok=True)
    np.concatenate([mie, chopped]).astype("<f4").tofile(OUT)
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")
    print("mie[0], mie[4095], chopped[4095] =", mie[0], mie[4095], chopped[4095])
    print("differing entries:", np.nonzero(mie != chopped)[0])
    return 0


if __name__ == "__main__":
    sys.exit(main())
