#!/bin/bash
# A/B PMC comparison of kernel variants selected by environment variables.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/pmc_ab
rm -rf "$OUT"; mkdir -p "$OUT"
run() {  # $1 tag, rest = env assignments
  tag=$1; shift
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
    n=$((n+1))
    env "$@" timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d "$OUT/$tag$n" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/$tag$n.log" 2>&1 || echo "$tag$n failed"
  done
}
n=0; run base CT_DUMMY=1
n=0; run pool CT_POOL=1
python3 - <<PY
import csv, glob, collections
for tag in ("base","pool"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s*/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_p" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", tag)
    for k in sorted(agg):
        v=agg[k]; print("  %-28s %16.4g (n=%d, last=%.4g)" % (k, sum(v)/len(v), len(v), v[-1]))
PY
