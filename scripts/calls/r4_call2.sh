#!/bin/bash
# round 4, call 2: GPU suite (without the full-size file) on the wide-node chains; A/B chains vs range-minimum tables vs round-3 library;
# instruction / wait attribution of the stream kernel's phases (timing-only builds)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call2; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
REPS=5 scripts/ab.sh base widermq r3 > $out/ab_c3.txt 2>&1; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base widermq r3 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base widermq r3 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
scripts/exp_pmc.sh $out/pmc base scanmin skipgather skip2a2c skip2c > $out/pmc.txt 2>&1; echo "[pmc]"; cat $out/pmc.txt
