#!/bin/bash
# Round 4: one re-sweep of the schedule knobs that the fused scheduler iterations could have moved (both kernels).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${1:-r04m}; mkdir -p "$OUT"; LOG="$OUT/knob_resweep.log"; : > "$LOG"
run() {  # estimator label env...
  local est=$1 label=$2; shift 2
  env "$@" python bench.py --estimator $est --steps ${STEPS:-4} --no-cpu-baseline --no-pmc-traffic --no-delta-leg --no-progressive-leg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('est $est $label:', round(d['value'], 1), 'Msamples/s', round(d['roofline']['avg_launch_ms'], 2), 'ms')" | tee -a "$LOG"
}
run 1 "defaults" CT_DUMMY=1
run 1 "regen_min 2" CT_REGEN_MIN=2
run 1 "regen_min 8" CT_REGEN_MIN=8
run 1 "regen_min 16" CT_REGEN_MIN=16
run 1 "burst_scatter 32" CT_BURST_SCATTER=32
run 1 "burst_scatter 64" CT_BURST_SCATTER=64
run 1 "march_burst 1" CT_MARCH_BURST=1
run 1 "defaults" CT_DUMMY=1
run 0 "defaults" CT_DUMMY=1
run 0 "scatter_min 4" CT_SCATTER_MIN=4
run 0 "scatter_min 8" CT_SCATTER_MIN=8
run 0 "burst_scatter 32" CT_BURST_SCATTER=32
run 0 "burst_scatter 48" CT_BURST_SCATTER=48
run 0 "march_burst 6" CT_MARCH_BURST=6
run 0 "march_burst 12" CT_MARCH_BURST=12
run 0 "regen_min 16" CT_REGEN_MIN=16
run 0 "defaults" CT_DUMMY=1
