#!/bin/bash
# round 4, call 17: only the lanes that hold rows of the task take part in the DMA of its first and last chunk — suite + A/B + request counts
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call17; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
REPS=7 scripts/ab.sh base nomaskdma > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--hits-per-query 10 --queries 20000000" REPS=3 scripts/ab.sh base nomaskdma > $out/ab_h10.txt 2>&1; echo "[10 hits]"; cat $out/ab_h10.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base nomaskdma > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
scripts/pmc_variants.sh base nomaskdma > $out/pmcv.txt 2>&1; echo "[requests]"; tail -3 $out/pmcv.txt
