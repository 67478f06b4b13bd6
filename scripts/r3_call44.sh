#!/bin/bash
# round-3 GPU call 44: node-id read-back in the kernel without the ring (its packed build has 60 B/lane of scratch) — C5 and uniform 1..N tables
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=4 AB_ARGS="--config C5" scripts/ab.sh base nr_reload > gpurun_out/c44_c5.log 2>&1; cat gpurun_out/c44_c5.log
timeout -k 10 300 python3 scripts/mixed_bench.py 2>&1 | grep uniform > gpurun_out/c44_mixed_base.log; cat gpurun_out/c44_mixed_base.log
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_nr_reload.so timeout -k 10 300 python3 scripts/mixed_bench.py 2>&1 | grep uniform > gpurun_out/c44_mixed_nr.log; cat gpurun_out/c44_mixed_nr.log
