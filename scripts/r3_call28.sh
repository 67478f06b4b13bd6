#!/bin/bash
# round-3 GPU call 28: attribution of the C5 stream kernel by timing-only builds (records are wrong in those)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=2 AB_ARGS="--config C5" scripts/ab.sh base no2c nolevels nogather > gpurun_out/c28_ab.log 2>&1; cat gpurun_out/c28_ab.log
