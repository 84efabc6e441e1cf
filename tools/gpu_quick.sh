#!/bin/bash
# Quick GPU iteration: parity tests (optionally filtered) + bench line.
set -o pipefail
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
rc=$?; echo "pytest exit $rc" >> gpurun_out/pytest_gpu.log; tail -15 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps ${STEPS:-2} --warmup 1 ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
