#!/bin/bash
# round 4, call 15: a dense step asks for the list entries' lines before its own record loads — suite (fast part) + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call15; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
AB_ARGS="--top-group zymo" REPS=7 scripts/ab.sh base nopf > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
REPS=5 scripts/ab.sh base nopf > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group all --queries 2000000" REPS=3 scripts/ab.sh base nopf > $out/ab_all.txt 2>&1; echo "[all tied]"; cat $out/ab_all.txt
AB_ARGS="--top-group zymo --strategy cautious" REPS=3 scripts/ab.sh base nopf > $out/ab_zc.txt 2>&1; echo "[zymo cautious]"; cat $out/ab_zc.txt
python3 scripts/ties_bench.py > $out/ties.txt 2>&1; echo "[ties]"; tail -8 $out/ties.txt
