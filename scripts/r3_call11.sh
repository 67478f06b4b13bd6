#!/bin/bash
# round 3, GPU call 11: flat phase-1 pass of the kernel without the ring: tests (full-size C5 included), C5 A/B against the build before it
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c11; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -5 $out/tests.txt
(REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
(REPS=2 scripts/ab.sh prev base) > $out/ab_c3.txt 2>&1; grep median $out/ab_c3.txt
python3 scripts/mixed_bench.py > $out/mixed.txt 2>&1; tail -8 $out/mixed.txt
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_prev.so python3 scripts/mixed_bench.py > $out/mixed_prev.txt 2>&1; tail -8 $out/mixed_prev.txt
