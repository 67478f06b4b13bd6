#!/bin/bash
# round 4, call 46: the ingest's device-wide prefix sums written in the file (three launches) instead of rocPRIM's: ingest + pipeline + C-ABI tests, end to end
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call46; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; rc=$?; echo "[tests] rc=$rc"; tail -3 $out/tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|line index|query: names|distinct to host|engine" $out/e2e.txt | head -30; tail -1 $out/e2e.txt
