#!/bin/bash
# Round 4, call g: why the march bricks behind virtual memory are slow -- the padded rows alone (CT_SPARSE=3, ordinary memory),
# the mapping alone (CT_SPARSE=4: 2-MiB chunks, no padding, nothing shared), both (2), neither (0); then the default bench.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04g; mkdir -p "$OUT"; : > "$OUT/summary.log"
for S in 0 3 4 2; do
  CT_SPARSE=$S CT_SCRATCH_GIB=32 timeout -k 10 500 python bench.py --volume 1024 --width 2048 --height 2048 --spp-per-step 512 --steps 2 --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2> "$OUT/bench_c4_sparse$S.err" > "$OUT/bench_c4_sparse$S.json" || { echo "bench CT_SPARSE=$S failed"; tail -5 "$OUT/bench_c4_sparse$S.err"; continue; }
  python - "$OUT/bench_c4_sparse$S.json" $S <<'PY' | tee -a "$OUT/summary.log"
import json, sys
d = json.load(open(sys.argv[1]))
m = d["config"]["volume_memory"]
print("1024^3 MARCH CT_SPARSE", sys.argv[2], round(d["value"], 1), "Msamples/s", round(d["roofline"]["avg_launch_ms"], 2), "ms per launch; march bricks",
      round(m["march_bricks_stored"] / 1e9, 3), "GB stored of", round(m["march_bricks_dense"] / 1e9, 3), "GB; setup", round(d["setup_s"], 1), "s")
PY
done
timeout -k 10 400 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "default bench rc=$?" | tee -a "$OUT/summary.log"
python - "$OUT/bench_default.json" <<'PY' | tee -a "$OUT/summary.log"
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(round(d["value"], 1), r["frac"], json.dumps(r.get("hbm_split"))[:900])
print("sq", {k: v for k, v in (r.get("sq") or {}).items() if k != "definitions"}); print("ea", r.get("ea"))
print("latency", {k: v for k, v in r["latency_model"].items() if k.startswith("frac")})
de = d["delta_estimator"]; print("delta", round(de["value"], 1), de["roofline"].get("traffic_frac"))
PY
