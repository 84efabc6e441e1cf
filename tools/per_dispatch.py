#!/usr/bin/env python3
"""Per-dispatch durations and PMC values of one kernel from a tools/gpu_profile.sh output directory.

    python tools/per_dispatch.py gpurun_out/prof_r01n render_persistent
"""
import csv, glob, os, sys
d, name = sys.argv[1], sys.argv[2]
f = max(glob.glob(d + "/trace/*/*_kernel_trace.csv"), key=os.path.getmtime)   # (a merged directory may hold older runs)
rows = [r for r in csv.DictReader(open(f)) if name in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("duration_ms", [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2) for r in rows])
for pd in sorted(glob.glob(d + "/pmc*/")):
    p = max(glob.glob(pd + "*/*_counter_collection.csv"), key=os.path.getmtime)
    by = {}
    for r in csv.DictReader(open(p)):
        if name in r["Kernel_Name"]:
            by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in sorted(by.items()):
        v.sort()
        print(k, [x[1] for x in v])
