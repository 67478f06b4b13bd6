#!/bin/bash
# round 3, GPU call 8: dense step whenever the list is full (zymo / C3 / all-tied), f64 layouts at 11 waves per CU with a 232-entry list
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c8; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
(REPS=3 AB_ARGS="--top-group zymo" scripts/ab.sh base densefall) > $out/ab_zymo.txt 2>&1; grep median $out/ab_zymo.txt
(REPS=3 scripts/ab.sh base densefall) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
(REPS=2 AB_ARGS="--top-group all --queries 2000000" scripts/ab.sh base densefall) > $out/ab_all.txt 2>&1; grep median $out/ab_all.txt
(REPS=2 AB_ARGS="--pident packed64" scripts/ab.sh base) > $out/ab_p64.txt 2>&1; grep median $out/ab_p64.txt
(REPS=2 AB_ARGS="--pident f64" scripts/ab.sh r2 base) > $out/ab_f64.txt 2>&1; grep median $out/ab_f64.txt
