#!/bin/bash
# round 4, call 11: priority defaults on (mode 4 + worklist finalisation); dense steps / ring requests at raised priority; whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call11; mkdir -p $out
REPS=5 scripts/ab.sh base noprio dense refill > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=5 scripts/ab.sh base noprio dense refill > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--top-group all --queries 2000000" REPS=3 scripts/ab.sh base noprio dense > $out/ab_all.txt 2>&1; echo "[all tied]"; cat $out/ab_all.txt
AB_ARGS="--config C5" REPS=5 scripts/ab.sh base noprio refill > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
