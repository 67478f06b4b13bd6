#!/bin/bash
# round 4, call 21: the staged parse kernel and the one-stream upload: ingest + pipeline tests, end to end (2 M queries) by
# the number of reader threads, kernel trace of one whole call
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call21; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
for t in 3 2 4; do
  BLU_UPLOAD_THREADS=$t timeout -k 10 300 python scripts/e2e_bench.py --reps 3 --dir /tmp/blu_e2e > $out/e2e_t$t.txt 2>&1; echo "[threads $t] rc=$?"
  grep -E "^rep|load db|start-up|upload|line index|parse |dictionary|engine|render  |writer  " $out/e2e_t$t.txt; tail -1 $out/e2e_t$t.txt
done
scripts/pipeline_profile.sh r14 2000000 > $out/profile.txt 2>&1; echo "[profile] rc=$?"; tail -45 $out/profile.txt
