#!/bin/bash
# round 4, call 22: parse kernel with its loads batched; tear-down laps; kernel trace of one whole call
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call22; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 4 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|load db|start-up|upload|line index|parse |dictionary|engine|render  |writer  |hand-over|free the|tear-down|load hits" $out/e2e.txt; tail -1 $out/e2e.txt
scripts/pipeline_profile.sh r14 2000000 > $out/profile.txt 2>&1; echo "[profile] rc=$?"
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pipeline_r14/kernel_stats.csv')))
print("total kernels ms", sum(int(r['TotalDurationNs']) for r in rows)/1e6)
for r in rows[:8]: print(r['Name'][:70], r['Calls'], int(r['TotalDurationNs'])/1e6)
PY
