#!/bin/bash
# round-3 GPU call 42: node-id read-back per layout — GPU suite, then A/B per layout (prev = the build before, base = f64 side records only, nl_all = f64 / milli columns too)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c42_tests.log 2>&1 || { tail -40 gpurun_out/c42_tests.log; exit 1; }
tail -3 gpurun_out/c42_tests.log
REPS=3 AB_ARGS="--pident packed64" scripts/ab.sh prev base > gpurun_out/c42_p64.log 2>&1; cat gpurun_out/c42_p64.log
REPS=3 AB_ARGS="--pident f64" scripts/ab.sh base nl_all > gpurun_out/c42_f64.log 2>&1; cat gpurun_out/c42_f64.log
REPS=3 AB_ARGS="--pident milli" scripts/ab.sh base nl_all > gpurun_out/c42_milli.log 2>&1; cat gpurun_out/c42_milli.log
REPS=3 scripts/ab.sh prev base > gpurun_out/c42_c3.log 2>&1; cat gpurun_out/c42_c3.log
