#!/bin/bash
# bench.py at BASELINE.json configs[1] (256^3, 512^2; steps of 1024 spp = four 256-spp jobs, so that a launch is long enough to be compared with the other sizes) and configs[4] (1024^3, 2048^2, 512 spp per launch: 28 GB of per-sample scratch per region, CT_SCRATCH_GIB=32) on one GPU.
#   gpurun -- 'bash tools/gpu_configs_bench.sh <tag>'
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-configs}; mkdir -p "$OUT"
python bench.py --volume 256 --width 512 --height 512 --steps 8 --no-cpu-baseline > "$OUT/bench_c1.json" 2> "$OUT/bench_c1.err" || { tail -3 "$OUT/bench_c1.err"; exit 1; }
CT_SCRATCH_GIB=32 python bench.py --volume 1024 --width 2048 --height 2048 --spp-per-step 512 --steps 2 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err" || { tail -3 "$OUT/bench_c4.err"; exit 1; }
python - "$OUT" <<'PY'
import json, sys
for f in ("bench_c1", "bench_c4"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json"))
    r = d["roofline"]
    print(f, round(d["value"], 1), "Msamples/s; traffic frac", round(r["frac"], 3), "useful frac", round(r["useful_frac"], 3), "l2 hit", r.get("l2_hit_rate"), "launch ms", round(r["avg_launch_ms"], 1), "DELTA", round(d.get("delta_estimator", {}).get("value", 0), 1), "progressive", d.get("progressive_10spp", {}).get("value"))
PY
