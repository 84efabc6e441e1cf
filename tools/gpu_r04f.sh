#!/bin/bash
# Round 4, call f: march bricks behind virtual memory (CT_SPARSE=2) -- parity, then configs[4] dense / virtual / row extents.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04f; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -s -k "vmm or scheduler_knobs or delta_fetch_layouts or checkpoint" > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee "$OUT/summary.log"; tail -4 "$OUT/tests.log" | tee -a "$OUT/summary.log"
[ "$(grep -c 'passed' "$OUT/tests.log")" -ge 1 ] || { tail -40 "$OUT/tests.log"; exit 1; }
grep -q "failed" "$OUT/tests.log" && { grep -B30 "Error" "$OUT/tests.log" | tail -60; exit 1; }
for S in 0 2 1 0 2; do
  CT_SPARSE=$S CT_SCRATCH_GIB=32 timeout -k 10 500 python bench.py --volume 1024 --width 2048 --height 2048 --spp-per-step 512 --steps 2 --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2> "$OUT/bench_c4_sparse$S.err" > "$OUT/bench_c4_sparse$S.json" || { echo "bench CT_SPARSE=$S failed"; tail -5 "$OUT/bench_c4_sparse$S.err"; continue; }
  python - "$OUT/bench_c4_sparse$S.json" $S <<'PY' | tee -a "$OUT/summary.log"
import json, sys
d = json.load(open(sys.argv[1]))
m = d["config"]["volume_memory"]
print("1024^3 MARCH CT_SPARSE", sys.argv[2], round(d["value"], 1), "Msamples/s", round(d["roofline"]["avg_launch_ms"], 2), "ms per launch; march bricks",
      round(m["march_bricks_stored"] / 1e9, 3), "GB stored of", round(m["march_bricks_dense"] / 1e9, 3), "GB; fetches per sample", round(d["roofline"]["fetches_per_sample"], 2))
PY
done
timeout -k 10 400 python bench.py --steps 2 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "default bench rc=$?" | tee -a "$OUT/summary.log"
python - "$OUT/bench_default.json" <<'PY' | tee -a "$OUT/summary.log"
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(round(d["value"], 1), r["frac"], json.dumps(r.get("hbm_split"))[:1500])
print("sq", r.get("sq")); print("ea", r.get("ea"))
print("latency", {k: v for k, v in r["latency_model"].items() if k.startswith("frac")})
PY
