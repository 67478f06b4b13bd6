#!/bin/bash
# round-3 GPU call 49: cautious strategy on C3 (packed and f64 columns) with the node-id read-back in the cautious builds
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=3 AB_ARGS="--strategy cautious" scripts/ab.sh base caut > gpurun_out/c49_packed.log 2>&1; cat gpurun_out/c49_packed.log
REPS=2 AB_ARGS="--strategy cautious --pident f64" scripts/ab.sh base caut > gpurun_out/c49_f64.log 2>&1; cat gpurun_out/c49_f64.log
REPS=2 AB_ARGS="--strategy cautious --top-group zymo" scripts/ab.sh base caut > gpurun_out/c49_zymo.log 2>&1; cat gpurun_out/c49_zymo.log
