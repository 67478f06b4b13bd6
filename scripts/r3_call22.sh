#!/bin/bash
# round 3, GPU call 22: rank-code and cutoff lookups as LDS reads of their own (no generic-pointer loads): tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c22; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
(REPS=4 scripts/ab.sh prev base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh prev base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
(REPS=2 AB_ARGS="--pident f64" scripts/ab.sh prev base) > $out/ab_f64.txt 2>&1; grep median $out/ab_f64.txt
(REPS=2 AB_ARGS="--queries 1250000" scripts/ab.sh prev base) > $out/ab_slice.txt 2>&1; grep median $out/ab_slice.txt
