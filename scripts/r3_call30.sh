#!/bin/bash
# round-3 GPU call 30: the kernel without the ring at 12 waves per CU / 168 VGPRs instead of 16 / 128
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=3 AB_ARGS="--config C5" scripts/ab.sh base n12 > gpurun_out/c30_ab.log 2>&1; cat gpurun_out/c30_ab.log
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_n12.so timeout -k 10 300 python3 scripts/mixed_bench.py > gpurun_out/c30_mixed_n12.log 2>&1 && cat gpurun_out/c30_mixed_n12.log
