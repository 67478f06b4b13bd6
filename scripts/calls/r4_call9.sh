#!/bin/bash
# round 4, call 9: wave priority variants (1: level 2 from the gather to the row request; 2: level 3; 3: level 2 to the end of the task;
# 4: level 3, then 1 for the finalisation) + the full-size tests with the faithful-oracle windows
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call9; mkdir -p $out
REPS=7 scripts/ab.sh base prio prio2 prio3 prio4 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=5 scripts/ab.sh base prio prio2 prio3 prio4 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base prio prio2 prio3 prio4 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--config C5" REPS=3 scripts/ab.sh base prio prio3 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $out/tests_full.txt 2>&1; echo "[fullsize] rc=$?"; tail -3 $out/tests_full.txt
