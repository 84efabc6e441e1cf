#!/usr/bin/env python3
"""Table of hardware counters per TIMED full launch for the variants tools/gpu_pmc_variants.sh ran.

    python tools/pmc_variant_table.py gpurun_out/r04c render_delta

bench.py --steps 2 --warmup 1 dispatches the estimator kernel as: warm-up launch(es), a resume-only launch (fence), the two
timed launches, a resume-only launch; the two timed ones are rows[-3:-1] of the kernel's dispatches in every pass."""
import csv, glob, json, os, sys

out, kernel = sys.argv[1], sys.argv[2]
table = {}
for vdir in sorted(glob.glob(out + "/*/")):
    name = os.path.basename(vdir.rstrip("/"))
    vals, dur = {}, []
    for f in sorted(glob.glob(vdir + "pmc*/**/*counter_collection.csv", recursive=True)):
        per = {}
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for c, rows in per.items():
            rows.sort()
            timed = rows[-3:-1]
            vals[c] = sum(x[1] for x in timed) / max(len(timed), 1)
            dur.append(sum(x[2] for x in timed) / max(len(timed), 1) / 1e6)
    if not vals:
        continue
    v = dict(vals)
    v["launch_ms_under_pmc"] = sum(dur) / len(dur)
    # bench lines of the (profiled) runs: Msamples/s under the profiler, for orientation only
    for log in sorted(glob.glob(out + "/" + name + ".pmc2.log")):
        for l in open(log):
            if l.startswith("{") and '"roofline"' in l:
                v["Msamples_per_s_under_pmc"] = json.loads(l)["value"]
    table[name] = v
names = list(table)
derived = {
    "traffic_GB (FETCH_SIZE x2 + WRITE_SIZE)": lambda v: (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 1e9,
    "traffic_TBps": lambda v: (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 1e12 / (v["launch_ms_under_pmc"] * 1e-3),
    "L2_hit_rate": lambda v: v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1),
    "lane_occupancy (THREAD_CYCLES_VALU / 64 ACTIVE_INST_VALU)": lambda v: v.get("SQ_THREAD_CYCLES_VALU", 0) / max(64 * v.get("SQ_ACTIVE_INST_VALU", 0), 1),
    "wait_any / wave_cycles": lambda v: v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1),
    "wait_inst_any / wave_cycles": lambda v: v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 0), 1),
    "VALU issue frac of 2-cycle rate (1024 SIMDs, clock from GRBM)": lambda v: 2 * v.get("SQ_INSTS_VALU", 0) / 1024 / max(v.get("GRBM_GUI_ACTIVE", 0) / 8, 1),
    "clock_GHz (GRBM_GUI_ACTIVE / 8 / time)": lambda v: v.get("GRBM_GUI_ACTIVE", 0) / 8 / (v["launch_ms_under_pmc"] * 1e-3) / 1e9,
    "EA read bytes_GB (32B x 32 + rest x 64)": lambda v: (v.get("TCC_EA0_RDREQ_32B_sum", 0) * 32 + (v.get("TCC_EA0_RDREQ_sum", 0) - v.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64) / 1e9,
    "EA read to DRAM frac": lambda v: v.get("TCC_EA0_RDREQ_DRAM_sum", 0) / max(v.get("TCC_EA0_RDREQ_sum", 0), 1),
    "EA avg read latency (LEVEL / RDREQ, L2 cycles)": lambda v: v.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / max(v.get("TCC_EA0_RDREQ_sum", 0), 1),
}
keys = sorted({k for v in table.values() for k in v})
w = max(len(k) for k in list(derived) + keys) + 2
print(" " * w + "".join("%16s" % n for n in names))
for k in keys:
    print(k.ljust(w) + "".join("%16.5g" % table[n].get(k, float("nan")) for n in names))
print()
for k, f in derived.items():
    row = []
    for n in names:
        try:
            row.append("%16.4g" % f(table[n]))
        except Exception:
            row.append("%16s" % "-")
    print(k.ljust(w) + "".join(row))
