#!/bin/bash
# Round 4: is the MARCH kernel's traffic served by the Infinity Cache or by HBM?  The counters cannot split it, but the L2's
# memory-side latency counter (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ) can be read for the SAME kernel on volumes whose working set
# certainly fits the 256 MiB cache (256^3, 384^3) and certainly does not (640^3, 768^3): where does 512^3 sit between them?
#   per volume: working set of a 64-spp launch (tools/working_set.py) + one PMC pass of bench.py's timed launches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04s}; mkdir -p "$OUT"; LOG="$OUT/ea_latency_by_volume.txt"; : > "$LOG"
for V in ${VOLUMES:-256 384 512 640 768}; do
  CT_STATS=1 timeout -k 10 300 python tools/working_set.py --volume $V --estimator ${EST:-0} --spp 64 2>/dev/null | grep touched_MiB > "$OUT/ws_$V.json"
  timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_$V" -- \
    python3 bench.py --volume $V --estimator ${EST:-0} --steps 2 --warmup 1 --no-cpu-baseline --no-delta-leg --no-progressive-leg --no-pmc-traffic > "$OUT/pmc_$V.log" 2>&1 || { echo "pmc $V failed"; tail -3 "$OUT/pmc_$V.log"; continue; }
  python3 - "$OUT" $V <<'PY' | tee -a "$LOG"
import csv, glob, json, os, sys
out, V = sys.argv[1], sys.argv[2]
f = max(glob.glob(f"{out}/pmc_{V}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
per = {}
for r in csv.DictReader(open(f)):
    if "render_persistent_kernel" in r["Kernel_Name"] or "render_delta_kernel" in r["Kernel_Name"]:
        per.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
v = {}
for c, rows in per.items():
    rows.sort(); t = rows[-3:-1]
    v[c] = sum(x[1] for x in t) / len(t); ms = sum(x[2] for x in t) / len(t) / 1e6
ws = json.loads(open(f"{out}/ws_{V}.json").read().strip().splitlines()[-1])["touched_MiB"] if os.path.getsize(f"{out}/ws_{V}.json") else float("nan")
line = [l for l in open(f"{out}/pmc_{V}.log") if l.startswith("{") and '"roofline"' in l]
val = json.loads(line[-1])["value"] if line else float("nan")
print(f"volume {V}^3: working set {ws:7.1f} MiB | launch {ms:7.2f} ms under pmc, {val:7.1f} Msamples/s | L2 misses {v['TCC_MISS_sum']:.3e} ({v['TCC_MISS_sum'] * 128 / 1e9 / (ms * 1e-3) / 1e3:.2f} TB/s), hit rate {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f} | fabric read latency {v['TCC_EA0_RDREQ_LEVEL_sum'] / v['TCC_EA0_RDREQ_sum']:7.1f} L2 cycles")
PY
  find "$OUT/pmc_$V" -name "*.csv" -size +1M -delete
done
