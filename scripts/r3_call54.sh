#!/bin/bash
# round-3 GPU call 54: cache-policy bits on the worklist kernel's bit-score loads (segments kept in registers: nt / sc1 nt; pass 1 of longer ones: nt) on C5 and on 600-hit queries
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=3 AB_ARGS="--config C5" scripts/ab.sh base lk2 lk18 lk2p2 > gpurun_out/c54_c5.log 2>&1; cat gpurun_out/c54_c5.log
REPS=2 AB_ARGS="--queries 800000 --hits-per-query 600" scripts/ab.sh base lk2 lk2p2 > gpurun_out/c54_600.log 2>&1; cat gpurun_out/c54_600.log
REPS=2 AB_ARGS="--queries 160000 --hits-per-query 3000" scripts/ab.sh base lk2p2 > gpurun_out/c54_3000.log 2>&1; cat gpurun_out/c54_3000.log
