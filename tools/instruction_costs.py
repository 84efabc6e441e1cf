#!/usr/bin/env python3
"""Static instruction counts (gfx950 ISA) of the pieces of a bounce, from tools/probes/cost_probe.hip: one tiny kernel per
piece, compiled with the product's flags; needs hipcc only (no GPU).  python tools/instruction_costs.py"""
import subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
with tempfile.TemporaryDirectory() as d:
    out = Path(d) / "cost.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", str(out), "cost_probe.hip"], check=True,
                   cwd=str(ROOT / "tools" / "probes"), stderr=subprocess.DEVNULL)
    s = out.read_text()
base = None
for name in ("k_empty", "k_newdir", "k_costheta", "k_sincos", "k_log", "k_exp", "k_div", "k_rcp", "k_sqrt", "k_rcpm", "k_sqrtm", "k_tea",
             "k_filter", "k_fetchm", "k_inbox"):
    i = s.find("\n" + name + ":")
    body = s[i:s.find(".Lfunc_end", i)]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    valu = sum(x.startswith("v_") for x in ins)
    base = valu if base is None else base
    print(f"{name[2:]:10s} VALU {valu - base:4d}   (all instructions {len(ins)})")
