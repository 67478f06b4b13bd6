#!/bin/bash
# round-3 GPU call 46: bit-score steps in flight in the flat pass (4 / 3 / 2) on C5
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=4 AB_ARGS="--config C5" scripts/ab.sh base d3 d2 > gpurun_out/c46_c5.log 2>&1; cat gpurun_out/c46_c5.log
