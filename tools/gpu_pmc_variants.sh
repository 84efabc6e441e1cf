#!/bin/bash
# Round 4: hardware counters of the timed launches of bench.py for several builds / settings of one estimator kernel.
#   VARIANTS="name:ENV=V,ENV2=V ..."  (CT_LIBRARY=libcloudtrace_w8.so selects another in-tree build)
#   ARGS="--estimator 1"              bench arguments;  KERNEL=render_delta   kernel name substring
# Four passes per variant (SQ; FETCH_SIZE; WRITE_SIZE + L2 hit/miss; the L2's memory-side request counters with the
# in-flight level whose quotient is the average fabric read latency).  Table: gpurun_out/<tag>/pmc_variants.txt
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r04c}; mkdir -p "$OUT"
ARGS=${ARGS:---estimator 1}
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-delta-leg --no-progressive-leg --no-pmc-traffic $ARGS"
for V in ${VARIANTS:-nee0:CT_DELTA_NEE=0 nee1:CT_DELTA_NEE=1 nee2:CT_DELTA_NEE=2}; do
  name=${V%%:*}; envs=${V#*:}
  i=0
  for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "FETCH_SIZE" \
             "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    ( export ${envs//,/ }; timeout -k 10 400 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/$name/pmc$i" -- $BENCH > "$OUT/$name.pmc$i.log" 2>&1 ) || { echo "$name pmc$i failed"; tail -3 "$OUT/$name.pmc$i.log"; }
  done
done
python3 tools/pmc_variant_table.py "$OUT" "${KERNEL:-render_delta}" | tee "$OUT/pmc_variants.txt"
# keep the counter CSVs out of the merge (large); the table and the bench lines stay
find "$OUT" -name "*.csv" -size +2M -delete 2>/dev/null; true
