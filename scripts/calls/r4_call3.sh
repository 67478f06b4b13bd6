#!/bin/bash
# round 4, call 3: after the sc1 pushes replaced the kernel-end release fence — chains vs range-minimum tables vs round 3;
# is the stream kernel bound by instruction issue? (padding instructions; two waves per SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call3; mkdir -p $out
REPS=5 scripts/ab.sh base widermq r3 pad200 pad400 w8 > $out/ab_c3.txt 2>&1; cat $out/ab_c3.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base widermq r3 w8 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh base widermq r3 > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
AB_ARGS="--config C5" REPS=3 scripts/ab.sh base r3 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
