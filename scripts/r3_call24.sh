#!/bin/bash
# round-3 GPU call 24: phase stamps of the stream kernel on C5 (no-ring build) and on the zymo-like C3 table
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so
timeout -k 10 300 python3 scripts/stamps.py --config C5 > gpurun_out/c24_stamps_c5.log 2>&1 && cat gpurun_out/c24_stamps_c5.log
timeout -k 10 300 python3 scripts/stamps.py --config C3 --top-group zymo > gpurun_out/c24_stamps_zymo.log 2>&1 && cat gpurun_out/c24_stamps_zymo.log
