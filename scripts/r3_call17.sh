#!/bin/bash
# round 3, GPU call 17: end to end with 6 / 10 / 14 upload threads
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 6 10 14; do
  echo "== BLU_UPLOAD_THREADS=$n"
  BLU_UPLOAD_THREADS=$n timeout -k 10 300 python3 scripts/e2e_bench.py --reps 3 2>&1 | grep -E "rep |upload text|device start-up|e2e_mqps"
done
nproc
