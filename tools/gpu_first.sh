#!/bin/bash
# First GPU round: parity tests, smoke, a short bench, a kernel-trace profile.
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/nproc.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_full_size_properties_512_1024 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout -k 10 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -5 gpurun_out/bench.log
