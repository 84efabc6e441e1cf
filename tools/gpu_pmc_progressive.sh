#!/bin/bash
# Per-launch counters of the estimator kernel at the reference's cadence (10 subframes per launch) and for one long batch.
#   [BENCH_ARGS="--spp 10 --updates 24 --ahead 80 --reference-spp 500"] tools/gpu_pmc_progressive.sh OUT_DIR [ENV...]
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$((n+1))
  env "$@" timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$n" -- python3 tools/progressive_bench.py ${BENCH_ARGS:---spp 10 --updates 24 --reference-spp 500} > "$OUT/p$n.log" 2>&1 || echo "pass $n failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_persistent" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
# dispatch ids differ between passes: match by order
ids = sorted(acc)
rows = [acc[i] for i in ids]
def cls(c):
    v = c.get("SQ_INSTS_VALU") or 0
    return v
print("dispatches:", len(rows))
for i, c in zip(ids, rows):
    print(i, " ".join("%s=%.4g" % (k.replace("SQ_", "").replace("_sum", ""), v) for k, v in sorted(c.items())))
PY
