#!/bin/bash
# round 4, call 24: accession sort beside the query dictionary, string vectors sized by the strings thread, tear-down laps:
# ingest + pipeline + C-ABI tests, end to end (2 M queries)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call24; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 4 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|load db|start-up|upload|line index|parse |dictionary|acc:|engine|render  |writer  |hand-over|free|tear-down|load hits|strings" $out/e2e.txt; tail -1 $out/e2e.txt
