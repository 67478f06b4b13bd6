#!/bin/bash
# round 4, call 5: tail pieces (all waves get a piece of the last partial round) — suite + A/B; the line-mix probe (memory-system
# ceiling of the C3 traffic with and without the dependency chain); the default bench line (pack_ms, ceiling, secondary)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call5; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/probe/mix.py > $out/mix.txt 2>&1; echo "[mix]"; cat $out/mix.txt
AB_ARGS="--queries 1250000" REPS=5 scripts/ab.sh base notail r3 > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
REPS=3 scripts/ab.sh base notail r3 > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--config C2" REPS=3 scripts/ab.sh base notail r3 > $out/ab_c2.txt 2>&1; echo "[c2]"; cat $out/ab_c2.txt
AB_ARGS="--config C5" REPS=3 scripts/ab.sh base notail r3 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.log; echo "[bench] rc=$?"; tail -14 $out/bench.log
