#!/bin/bash
# round-3 GPU call 25: flat pass with deferred gather — GPU suite, then A/B on C5 and on uniform 1..N tables
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c25_tests.log 2>&1 || { tail -40 gpurun_out/c25_tests.log; exit 1; }
tail -3 gpurun_out/c25_tests.log
REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base > gpurun_out/c25_ab_c5.log 2>&1 && cat gpurun_out/c25_ab_c5.log
timeout -k 10 300 python3 scripts/mixed_bench.py > gpurun_out/c25_mixed_base.log 2>&1 && cat gpurun_out/c25_mixed_base.log
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_prev.so timeout -k 10 300 python3 scripts/mixed_bench.py > gpurun_out/c25_mixed_prev.log 2>&1 && cat gpurun_out/c25_mixed_prev.log
