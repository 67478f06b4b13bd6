#!/bin/bash
# round-3 GPU call 31: flat pass in two sub-passes — GPU suite, then A/B on C5 and the uniform 1..N tables
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c31_tests.log 2>&1 || { tail -40 gpurun_out/c31_tests.log; exit 1; }
tail -3 gpurun_out/c31_tests.log
REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base d2 n12 > gpurun_out/c31_ab_c5.log 2>&1 && cat gpurun_out/c31_ab_c5.log
timeout -k 10 300 python3 scripts/mixed_bench.py > gpurun_out/c31_mixed_base.log 2>&1 && cat gpurun_out/c31_mixed_base.log
