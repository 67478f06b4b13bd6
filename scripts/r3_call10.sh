#!/bin/bash
# round 3, GPU call 10: the whole GPU suite (full-size tests included), smoke, default bench with the secondary workloads
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c10; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err; tail -12 $out/bench.err; cat $out/bench.json | head -c 3000
