#!/bin/bash
# round-3 GPU call 48: the cautious strategy next to the relaxed one on C3 (its packed ring build has 12 B/lane of scratch)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=3 scripts/ab.sh base > gpurun_out/c48_relaxed.log 2>&1; cat gpurun_out/c48_relaxed.log
REPS=3 AB_ARGS="--strategy cautious" scripts/ab.sh base > gpurun_out/c48_cautious.log 2>&1; cat gpurun_out/c48_cautious.log
