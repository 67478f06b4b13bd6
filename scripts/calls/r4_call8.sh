#!/bin/bash
# round 4, call 8: raised wave priority on the dependency chain gather -> phase 2a -> row request; wide-node block entry requested at the end of phase 2a
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call8; mkdir -p $out
REPS=7 scripts/ab.sh base prio wearly both r3 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=5 scripts/ab.sh base prio wearly both > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base prio wearly both > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--config C5" REPS=3 scripts/ab.sh base prio > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
