#!/bin/bash
# round 4, call 31: worklist kernel at 7 waves per SIMD — the whole GPU suite, then final profiles part 1 (C3 packed, C5: kernel trace + PMC; zymo-like: trace)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4_call31
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_call31/tests.txt 2>&1; rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/r4_call31/tests.txt
[ $rc -eq 0 ] || exit 1
scripts/final_profiles.sh r15 "c3:C3-packed:" "c5:C5-packed:--config C5" "zymo:-:--top-group zymo"
ls gpurun_out | grep profiles_r15
