#!/bin/bash
# round-3 GPU call 38: shape hint carried through the dense steps — GPU suite, A/B on the zymo-like table and on C3
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c38_tests.log 2>&1 || { tail -40 gpurun_out/c38_tests.log; exit 1; }
tail -3 gpurun_out/c38_tests.log
REPS=3 AB_ARGS="--top-group zymo" scripts/ab.sh prev base > gpurun_out/c38_ab_zymo.log 2>&1 && cat gpurun_out/c38_ab_zymo.log
REPS=3 scripts/ab.sh prev base > gpurun_out/c38_ab_c3.log 2>&1 && cat gpurun_out/c38_ab_c3.log
