#!/bin/bash
# round-3 GPU call 23: GPU suite (without the full-size file) on the division-free milli conversion, then the same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c23_tests.log 2>&1 || { tail -30 gpurun_out/c23_tests.log; exit 1; }
tail -3 gpurun_out/c23_tests.log
REPS=3 scripts/ab.sh prev base > gpurun_out/c23_ab_c3.log 2>&1 && cat gpurun_out/c23_ab_c3.log
REPS=3 AB_ARGS="--pident packed64" scripts/ab.sh prev base > gpurun_out/c23_ab_p64.log 2>&1 && cat gpurun_out/c23_ab_p64.log
REPS=3 AB_ARGS="--pident f64" scripts/ab.sh prev base > gpurun_out/c23_ab_f64.log 2>&1 && cat gpurun_out/c23_ab_f64.log
