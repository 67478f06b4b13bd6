#!/bin/bash
# round-3 GPU call 51: node ids of the first 9 / 12 levels in registers (the rest read back) against all 20, packed ring build
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=4 scripts/ab.sh base nid9 nid12 > gpurun_out/c51_c3.log 2>&1; cat gpurun_out/c51_c3.log
REPS=3 AB_ARGS="--top-group zymo" scripts/ab.sh base nid9 > gpurun_out/c51_zymo.log 2>&1; cat gpurun_out/c51_zymo.log
REPS=3 AB_ARGS="--queries 1250000" scripts/ab.sh base nid9 > gpurun_out/c51_slice.log 2>&1; cat gpurun_out/c51_slice.log
