#!/bin/bash
# round 4, call 26: taxonomy tables in one device allocation, call memory released by a thread: the whole GPU suite, end to end
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call26; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|start-up|upload|taxonomy|tear-down" $out/e2e.txt; tail -1 $out/e2e.txt
