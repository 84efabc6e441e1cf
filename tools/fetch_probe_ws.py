#!/usr/bin/env python3
"""Random-128-B-line fill rate by WORKING-SET size (round 4; ct_debug_fetch_probe_ws): the estimator's access shape -- two
unaligned 8-byte loads from one pseudo-random line per lane -- over sets of 2 MiB ... 4 GiB.  A set that fits a cache level
is re-read from that level, so the rates are the line-fill ceilings of L2 (4 MiB per XCD), the Infinity Cache (256 MiB) and
HBM for this access shape.

    python tools/fetch_probe_ws.py                 timing (2^25 lanes per launch; rate from the difference of 12 and 4 launches)
    rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d DIR -- \\
        python3 tools/fetch_probe_ws.py --pmc      one launch group per size, in the order printed: per-size counters
    python tools/fetch_probe_ws.py --table DIR     the per-size table from that run's counter_collection.csv
"""
import argparse, csv, ctypes as C, glob, json, os, sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
SIZES_MIB = [2, 8, 32, 64, 128, 192, 256, 384, 512, 1024, 4096]
LOG2_THREADS = 25
PMC_REPEATS = 6


def table(d):
    f = max(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    per = {}
    for r in csv.DictReader(open(f)):
        if "fetch_probe_ws_kernel" in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
            per[int(r["Dispatch_Id"])]["ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(per)
    assert len(ids) == PMC_REPEATS * len(SIZES_MIB), (len(ids), PMC_REPEATS * len(SIZES_MIB))
    print("%8s %12s %12s %10s %14s %12s" % ("MiB", "EA reads", "EA lat (cyc)", "L2 hit", "lines/s under pmc", "GB/s"))
    out = []
    for k, mib in enumerate(SIZES_MIB):
        rows = [per[i] for i in ids[k * PMC_REPEATS + 2:(k + 1) * PMC_REPEATS]]     # (the first two launches warm the level up)
        rd = sum(r.get("TCC_EA0_RDREQ_sum", 0) for r in rows) / len(rows)
        lv = sum(r.get("TCC_EA0_RDREQ_LEVEL_sum", 0) for r in rows) / len(rows)
        hit = sum(r.get("TCC_HIT_sum", 0) for r in rows)
        miss = sum(r.get("TCC_MISS_sum", 0) for r in rows)
        ns = sum(r["ns"] for r in rows) / len(rows)
        rate = (1 << LOG2_THREADS) / (ns * 1e-9)
        rec = {"working_set_MiB": mib, "ea_read_requests_per_launch": rd, "ea_avg_read_latency_cycles": lv / max(rd, 1),
               "l2_hit_rate": hit / max(hit + miss, 1), "lines_per_s_under_pmc": rate}
        out.append(rec)
        print("%8d %12.4g %12.1f %10.3f %14.4g %12.1f" % (mib, rd, rec["ea_avg_read_latency_cycles"], rec["l2_hit_rate"], rate, rate * 128 / 1e9))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--table", default=None)
    a = ap.parse_args()
    if a.table:
        return table(a.table)
    from deepestscatter_amd import _lib
    L = _lib.load()
    s = C.c_uint64(0)
    if a.pmc:
        for mib in SIZES_MIB:
            assert L.ct_debug_fetch_probe_ws(0, LOG2_THREADS, mib * 8192, PMC_REPEATS, C.byref(s)) == 0
            print("probed", mib, "MiB", flush=True)
        return
    out = []
    for mib in SIZES_MIB:
        t = {}
        for reps in (4, 12, 4, 12):
            t0 = time.perf_counter()
            assert L.ct_debug_fetch_probe_ws(0, LOG2_THREADS, mib * 8192, reps, C.byref(s)) == 0
            dt = time.perf_counter() - t0
            t[reps] = min(t.get(reps, dt), dt)
        per = (t[12] - t[4]) / 8.0
        rate = (1 << LOG2_THREADS) / per
        out.append({"working_set_MiB": mib, "lines_per_s": rate, "GBps": rate * 128 / 1e9})
        print("%6d MiB  %.3g lines/s  %.0f GB/s" % (mib, rate, rate * 128 / 1e9), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
