#!/bin/bash
# round-3 GPU call 37: worklist kernel whose waves take their queries off the queues — GPU suite, A/B on C5 and C3
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c37_tests.log 2>&1 || { tail -40 gpurun_out/c37_tests.log; exit 1; }
tail -3 gpurun_out/c37_tests.log
REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base > gpurun_out/c37_ab_c5.log 2>&1 && cat gpurun_out/c37_ab_c5.log
REPS=3 scripts/ab.sh prev base > gpurun_out/c37_ab_c3.log 2>&1 && cat gpurun_out/c37_ab_c3.log
