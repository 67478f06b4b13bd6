#!/bin/bash
# round 3, GPU call 19: 32-bit offset arithmetic in the stream kernel: tests, A/B against the build before it
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c19; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -5 $out/tests.txt
(REPS=4 scripts/ab.sh prev base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
(REPS=2 AB_ARGS="--queries 1250000" scripts/ab.sh prev base) > $out/ab_slice.txt 2>&1; grep median $out/ab_slice.txt
(REPS=2 AB_ARGS="--hits-per-query 10 --queries 20000000" scripts/ab.sh prev base) > $out/ab_h10.txt 2>&1; grep median $out/ab_h10.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh prev base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
(REPS=2 AB_ARGS="--pident packed64" scripts/ab.sh prev base) > $out/ab_p64.txt 2>&1; grep median $out/ab_p64.txt
