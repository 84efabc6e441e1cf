#!/bin/bash
# Round 4, call d: new tests, working-set probe (timing + EA counters), touched-line working sets, DELTA layouts at 1024^3.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r04d; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_parity_gaps.py -q -m gpu -x -s -k "checkpoint or fixed_point or libm or delta" > "$OUT/tests.log" 2>&1; echo "tests rc=$?" | tee -a "$OUT/summary.log"; tail -3 "$OUT/tests.log" | tee -a "$OUT/summary.log"
grep "vs oracle" "$OUT/tests.log" | tee -a "$OUT/summary.log"
timeout -k 10 300 python tools/fetch_probe_ws.py > "$OUT/fetch_probe_ws.txt" 2>&1; tail -14 "$OUT/fetch_probe_ws.txt" | head -12
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/probe_pmc" -- python3 tools/fetch_probe_ws.py --pmc > "$OUT/probe_pmc.log" 2>&1 && python tools/fetch_probe_ws.py --table "$OUT/probe_pmc" | tee "$OUT/fetch_probe_ws_pmc.txt"
find "$OUT/probe_pmc" -name "*.csv" -size +1M -delete
CT_STATS=1 timeout -k 10 600 python tools/working_set.py --estimator 0 > "$OUT/working_set_march.jsonl" 2>&1; cat "$OUT/working_set_march.jsonl" | cut -c1-400
CT_STATS=1 CT_DELTA_NEE=1 timeout -k 10 600 python tools/working_set.py --estimator 1 > "$OUT/working_set_delta_nee1.jsonl" 2>&1; cut -c1-400 "$OUT/working_set_delta_nee1.jsonl"
CT_STATS=1 CT_DELTA_NEE=2 timeout -k 10 600 python tools/working_set.py --estimator 1 > "$OUT/working_set_delta_nee2.jsonl" 2>&1; cut -c1-400 "$OUT/working_set_delta_nee2.jsonl"
for NEE in 0 1 2; do
  CT_DELTA_NEE=$NEE CT_SCRATCH_GIB=32 timeout -k 10 500 python bench.py --volume 1024 --width 2048 --height 2048 --spp-per-step 512 --steps 2 --estimator 1 --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2>/dev/null > "$OUT/delta_1024_nee$NEE.json"
  python - "$OUT/delta_1024_nee$NEE.json" $NEE <<'PY' | tee -a "$OUT/summary.log"
import json, sys
d = json.load(open(sys.argv[1]))
print("1024^3 DELTA NEE", sys.argv[2], round(d["value"], 1), "Msamples/s", round(d["roofline"]["avg_launch_ms"], 2), "ms per launch")
PY
done
