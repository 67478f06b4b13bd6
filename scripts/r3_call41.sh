#!/bin/bash
# round-3 GPU call 41: the node id of the reported level read back from the reference row's line instead of kept in 20 registers
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=3 scripts/ab.sh base nodeload > gpurun_out/c41_c3.log 2>&1; cat gpurun_out/c41_c3.log
REPS=3 AB_ARGS="--top-group zymo" scripts/ab.sh base nodeload > gpurun_out/c41_zymo.log 2>&1; cat gpurun_out/c41_zymo.log
REPS=3 AB_ARGS="--pident packed64" scripts/ab.sh base nodeload > gpurun_out/c41_p64.log 2>&1; cat gpurun_out/c41_p64.log
REPS=3 AB_ARGS="--config C5" scripts/ab.sh base nodeload > gpurun_out/c41_c5.log 2>&1; cat gpurun_out/c41_c5.log
REPS=3 AB_ARGS="--queries 1250000" scripts/ab.sh base nodeload > gpurun_out/c41_slice.log 2>&1; cat gpurun_out/c41_slice.log
