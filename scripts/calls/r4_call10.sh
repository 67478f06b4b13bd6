#!/bin/bash
# round 4, call 10: priority 3 / 1 / 0 (mode 4) vs 3 / 2 / 0 (mode 5); the worklist kernel's finalisation at raised priority
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call10; mkdir -p $out
REPS=7 scripts/ab.sh base prio4 prio5 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=5 scripts/ab.sh base prio4 prio5 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--queries 1250000" REPS=7 scripts/ab.sh base prio4 prio5 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--config C5" REPS=7 scripts/ab.sh base prio4 prio4l > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
AB_ARGS="--pident packed64" REPS=3 scripts/ab.sh base prio4 > $out/ab_p64.txt 2>&1; echo "[p64]"; cat $out/ab_p64.txt
AB_ARGS="--pident f64" REPS=3 scripts/ab.sh base prio4 > $out/ab_f64.txt 2>&1; echo "[f64]"; cat $out/ab_f64.txt
AB_ARGS="--hits-per-query 10 --queries 20000000" REPS=3 scripts/ab.sh base prio4 > $out/ab_h10.txt 2>&1; echo "[10 hits]"; cat $out/ab_h10.txt
