#!/bin/bash
# Profiling round: kernel-trace stats + PMC passes of the bench command (bench without the CPU leg).
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_${1:-r01}
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-delta-leg --no-progressive-leg --no-pmc-traffic ${BENCH_EXTRA:-}"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc$i failed"; tail -5 "$OUT/pmc$i.log"; }
done
find "$OUT" -name "*.csv" | head -40
python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
