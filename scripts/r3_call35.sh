#!/bin/bash
# round-3 GPU call 35: C5 after the worklist queues — variants of the kernel without the ring and timing-only attribution
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=2 AB_ARGS="--config C5" scripts/ab.sh base n12 d2 empty no2c nop1 > gpurun_out/c35_ab.log 2>&1; cat gpurun_out/c35_ab.log
