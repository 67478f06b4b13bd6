#!/bin/bash
# round 4, call 18: the finalisation split in two halves (base = not deferred: must equal the profiled build's records and time);
# variant "defer": the second half of task i inside the gather window of task i + 1 (packed ring build) — parity suite on it, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call18; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $out/tests_base.txt 2>&1; echo "[tests base] rc=$?"; tail -2 $out/tests_base.txt
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_defer.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $out/tests_defer.txt 2>&1; echo "[tests defer] rc=$?"; tail -4 $out/tests_defer.txt
REPS=5 scripts/ab.sh base defer defer32 r3 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--queries 1250000" REPS=3 scripts/ab.sh base defer defer32 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base defer defer32 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
