#!/bin/bash
# A/B of the exchange kernel's schedule knobs on the benchmark scene (run on the GPU box through gpurun).
#   tools/gpu_exchange_ab.sh OUT_DIR "ENV1" "ENV2" ...     each ENV is a space-separated list of VAR=value
export CT_LIBRARY=libcloudtrace_exp.so   # the exchange kernels live in the experiments build (python -m deepestscatter_amd.build --variant exp)
out=$1; shift
mkdir -p "$out"
for cfg in "$@"; do
  echo "=== $cfg" >> "$out/ab.log"
  env $cfg timeout -k 10 200 python tools/exchange_check.py --skip-parity --steps 2 --spp 512 >> "$out/ab.log" 2>&1 || echo "FAILED rc=$?" >> "$out/ab.log"
done
grep -E "^===|^exchange |^per_lane|speedup|FAILED" "$out/ab.log"
