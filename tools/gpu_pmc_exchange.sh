#!/bin/bash
# PMC comparison of the per-lane DELTA kernel and the exchange kernel (same process, same job).  Usage (GPU box):
#   tools/gpu_pmc_exchange.sh OUT_DIR [ENV...]
export CT_LIBRARY=libcloudtrace_exp.so   # the exchange kernels live in the experiments build (python -m deepestscatter_amd.build --variant exp)
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=$1; shift
mkdir -p "$OUT"
n=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  n=$((n+1))
  env "$@" timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$n" -- python3 tools/exchange_check.py --skip-parity --steps 1 --spp 256 > "$OUT/p$n.log" 2>&1 || echo "pass $n failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tag = "exchange" if ("render_delta_x" in k or "render_delta_w" in k) else ("per_lane" if "render_delta_kernel" in k else None)
        if tag:
            agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag in ("per_lane", "exchange"):
    print("==", tag, "(last dispatch = the timed 256-spp launch)")
    a = agg[tag]
    for k in sorted(a):
        print("  %-26s %14.5g  (n=%d)" % (k, a[k][-1], len(a[k])))
    if "SQ_THREAD_CYCLES_VALU" in a and "SQ_ACTIVE_INST_VALU" in a:
        print("  lane occupancy of VALU instructions: %.3f" % (a["SQ_THREAD_CYCLES_VALU"][-1] / (64 * a["SQ_ACTIVE_INST_VALU"][-1])))
    if "SQ_WAIT_ANY" in a and "SQ_WAVE_CYCLES" in a:
        print("  SQ_WAIT_ANY / SQ_WAVE_CYCLES: %.3f" % (a["SQ_WAIT_ANY"][-1] / a["SQ_WAVE_CYCLES"][-1]))
PY
