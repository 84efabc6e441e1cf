#!/bin/bash
# Round 4, final evidence: quick parity of the changed kernel, smoke, the default bench line, then rocprofv3 kernel-trace +
# PMC passes of the bench command for both estimators (tools/gpu_profile.sh), per-dispatch lists.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04z; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "delta or knobs or differential" > "$OUT/tests_delta.log" 2>&1; echo "delta tests rc=$?"; tail -2 "$OUT/tests_delta.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
( time python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" ) 2>&1 | grep real; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("MARCH", round(d["value"], 1), "frac", round(r["frac"], 3), "launch ms", round(r["avg_launch_ms"], 2))
de = d["delta_estimator"]; print("DELTA", round(de["value"], 1), "traffic_frac", de["roofline"].get("traffic_frac"), "launch ms", round(de["avg_launch_ms"], 2))
print("progressive", round(d["progressive_10spp"]["value"], 1), "cpu", round(d["cpu_baseline"]["value"], 3), d["frame_sha256"][:16])
PY
bash tools/gpu_profile.sh r04z > "$OUT/profile_march.log" 2>&1; echo "profile march rc=$?"
BENCH_EXTRA="--estimator 1" bash tools/gpu_profile.sh r04z_delta > "$OUT/profile_delta.log" 2>&1; echo "profile delta rc=$?"
for t in r04z r04z_delta; do
  k=render_persistent; [ $t = r04z_delta ] && k=render_delta
  python tools/per_dispatch.py gpurun_out/prof_$t $k > gpurun_out/prof_$t/render_launches.txt 2>&1
  cp gpurun_out/prof_$t/trace/*/*kernel_stats.csv gpurun_out/prof_$t/kernel_stats.csv 2>/dev/null
  find gpurun_out/prof_$t -name "*.csv" -size +1M -delete; find gpurun_out/prof_$t -name "*_kernel_trace.csv" -delete
  head -3 gpurun_out/prof_$t/render_launches.txt | cut -c1-200
done
