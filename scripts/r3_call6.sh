#!/bin/bash
# round 3, GPU call 6: tests (without the full-size ones), then A/B of HEAD against the round-2 library on C3 / slice / C5 / C2
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c6; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -12 $out/tests.txt
(REPS=3 scripts/ab.sh r2 base) > $out/ab.txt 2>&1; grep median $out/ab.txt
(REPS=3 AB_ARGS="--queries 1250000" scripts/ab.sh r2 base) > $out/ab_slice.txt 2>&1; grep median $out/ab_slice.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh r2 base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
(REPS=2 AB_ARGS="--config C2" scripts/ab.sh r2 base) > $out/ab_c2.txt 2>&1; grep median $out/ab_c2.txt
