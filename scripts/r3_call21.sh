#!/bin/bash
# round 3, GPU call 21: phase 2a without the error tests in rounds that hold no row without a lineage: tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c21; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
(REPS=4 scripts/ab.sh prev base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
(REPS=2 AB_ARGS="--top-group zymo" scripts/ab.sh prev base) > $out/ab_zymo.txt 2>&1; grep median $out/ab_zymo.txt
(REPS=2 AB_ARGS="--pident packed64" scripts/ab.sh prev base) > $out/ab_p64.txt 2>&1; grep median $out/ab_p64.txt
