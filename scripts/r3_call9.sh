#!/bin/bash
# round 3, GPU call 9: overlapping lane windows in the ring scan (no per-row masking): tests, then A/B against the build before it
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c9; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
(REPS=4 scripts/ab.sh densefall base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
(REPS=2 AB_ARGS="--top-group zymo" scripts/ab.sh densefall base) > $out/ab_zymo.txt 2>&1; grep median $out/ab_zymo.txt
(REPS=2 AB_ARGS="--hits-per-query 100 --queries 5000000" scripts/ab.sh densefall base) > $out/ab_h100.txt 2>&1; grep median $out/ab_h100.txt
(REPS=2 AB_ARGS="--queries 1250000" scripts/ab.sh densefall base) > $out/ab_slice.txt 2>&1; grep median $out/ab_slice.txt
