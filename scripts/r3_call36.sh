#!/bin/bash
# round-3 GPU call 36: worklist kernel with level words + offsets one query ahead — GPU suite, A/B on C5 (8 / 7 / 6 waves per SIMD)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c36_tests.log 2>&1 || { tail -40 gpurun_out/c36_tests.log; exit 1; }
tail -3 gpurun_out/c36_tests.log
REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base b7 b6 > gpurun_out/c36_ab_c5.log 2>&1 && cat gpurun_out/c36_ab_c5.log
