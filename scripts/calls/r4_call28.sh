#!/bin/bash
# round 4, call 28: taxonomy handle released by the graveyard thread too: pipeline-side tests, end to end (2 M queries, 6 fresh processes)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call28; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py tests/test_gpu_golden_render.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 6 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|start-up|device buffer|upload text|tear-down|hand the" $out/e2e.txt; tail -1 $out/e2e.txt
