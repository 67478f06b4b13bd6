#!/bin/bash
# round 4, call 27: where the upload's and the tear-down's time goes (trace laps), with and without the taxonomy thread's company
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call27; mkdir -p $out
timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|start-up|upload|taxonomy|tear-down|small vectors|hand the" $out/e2e.txt; tail -1 $out/e2e.txt
