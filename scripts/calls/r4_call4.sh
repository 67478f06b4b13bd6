#!/bin/bash
# round 4, call 4: GPU suite (fast part) on the add-with-carry tie masks + the in-asm DMA request loop; A/B of each
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call4; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
REPS=5 scripts/ab.sh base nomask nodma neither r3 > $out/ab_c3.txt 2>&1; cat $out/ab_c3.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base neither r3 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--hits-per-query 10 --queries 20000000" REPS=3 scripts/ab.sh base neither > $out/ab_h10.txt 2>&1; echo "[10 hits]"; cat $out/ab_h10.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base neither > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
