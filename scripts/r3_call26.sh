#!/bin/bash
# round-3 GPU call 26: where C5's time is after the new flat pass — phase stamps and the per-kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so timeout -k 10 300 python3 scripts/stamps.py --config C5 > gpurun_out/c26_stamps_c5.log 2>&1 && cat gpurun_out/c26_stamps_c5.log
TRACE_ONLY=1 timeout -k 10 400 scripts/profile_round.sh c26_c5 --config C5 > gpurun_out/c26_trace.log 2>&1 && cat gpurun_out/c26_trace.log
