"""How many estimator kernels were resident at once: reads the kernel trace (CSV) under a rocprofv3 output directory.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/point_concurrency_probe.py 4
    python tools/kernel_overlap.py OUT
"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "render_" in r["Kernel_Name"]]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
busy = 0; conc_time = {}; cur = 0; last = t0
for t, d in ev:
    conc_time[cur] = conc_time.get(cur, 0) + (t - last); last = t; cur += d
durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("kernels", len(rows), "span ms %.1f" % ((t1 - t0) / 1e6), "sum ms %.1f" % (sum(durs) / 1e6), "median ms %.2f" % (durs[len(durs)//2] / 1e6))
print("time at concurrency:", {k: "%.1f ms" % (v / 1e6) for k, v in sorted(conc_time.items())})
