#!/bin/bash
# round 4, final check of the tree as committed: the whole GPU suite, smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_final; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "[smoke] rc=$?"; tail -1 $out/smoke.txt
BLU_BENCH_CEILING_TRACE=1 timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.log; echo "[bench] rc=$?"; tail -18 $out/bench.log
