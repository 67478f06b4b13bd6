#!/bin/bash
# round 4, call 29: pipeline-side tests; end to end with a pause between the fresh processes (0, 1, 3 s)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call29; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
for p in 0 1 3; do
  timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --pause $p --dir /tmp/blu_e2e > $out/e2e_p$p.txt 2>&1; echo "[pause $p] rc=$?"
  grep -E "start-up|device buffer" $out/e2e_p$p.txt | tr '\n' ' '; echo; tail -1 $out/e2e_p$p.txt
done
