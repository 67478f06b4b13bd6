#!/bin/bash
# round 4, call 6: status records put together at the staging point (no scratch in the packed ring build again) + tail pieces
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call6; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
REPS=5 scripts/ab.sh base notail r3 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base notail r3 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base r3 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--pident packed64" REPS=3 scripts/ab.sh base r3 > $out/ab_p64.txt 2>&1; echo "[p64]"; cat $out/ab_p64.txt
AB_ARGS="--config C5" REPS=3 scripts/ab.sh base r3 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
AB_ARGS="--strategy cautious" REPS=3 scripts/ab.sh base r3 > $out/ab_caut.txt 2>&1; echo "[cautious]"; cat $out/ab_caut.txt
