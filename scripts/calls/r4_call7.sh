#!/bin/bash
# round 4, call 7: lane laundering on/off, node ids kept in registers (12 / 16 / 20), vs round 3 — seven interleaved repetitions
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call7; mkdir -p $out
REPS=7 scripts/ab.sh base nolaunder nid16 nid20 r3 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=5 scripts/ab.sh base nolaunder nid20 r3 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base nolaunder nid20 r3 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--config C5" REPS=5 scripts/ab.sh base nolaunder r3 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
