#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small text/JSON summary."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
out = {}
print(f"# rocprofv3 summary of {root}")
for f in sorted(root.rglob("*kernel_stats.csv")):
    print(f"\n## kernel stats ({f.relative_to(root)})")
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  ", {k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
    out["kernel_stats"] = rows[:12]
per_kernel = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in sorted(root.rglob("*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "?").split("(")[0]
        per_kernel[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[name][r["Counter_Name"]] += 1
print("\n## PMC counters (mean per dispatch)")
pm = {}
for name, ctr in per_kernel.items():
    if "render" not in name and "accumulate" not in name:
        continue
    print(f"  {name}")
    pm[name] = {}
    for c, v in sorted(ctr.items()):
        mean = v / max(calls[name][c], 1)
        pm[name][c] = mean
        print(f"     {c:36s} {mean:18.1f}  (n={calls[name][c]})")
out["pmc_mean_per_dispatch"] = pm
(root / "summary.json").write_text(json.dumps(out, indent=1))
