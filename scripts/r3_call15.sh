#!/bin/bash
# round 3, GPU call 15: f64 layouts on comparison-ready entries when the identities are exact milli-percent values: tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c15; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -5 $out/tests.txt
(REPS=3 AB_ARGS="--pident packed64" scripts/ab.sh prev base) > $out/ab_p64.txt 2>&1; grep median $out/ab_p64.txt
(REPS=2 AB_ARGS="--pident f64" scripts/ab.sh prev base) > $out/ab_f64.txt 2>&1; grep median $out/ab_f64.txt
(REPS=2 scripts/ab.sh prev base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
