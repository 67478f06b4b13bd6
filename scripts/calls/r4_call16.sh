#!/bin/bash
# round 4, call 16: end to end (2 M queries, 6.7 GB of text) by the number of upload threads
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call16; mkdir -p $out
nproc > $out/nproc.txt; cat $out/nproc.txt
for t in 6 10 14; do
  BLU_UPLOAD_THREADS=$t timeout -k 10 300 python scripts/e2e_bench.py --reps 3 --dir /tmp/blu_e2e > $out/e2e_t$t.txt 2>&1; echo "[threads $t] rc=$?"; grep -E "^rep|upload|text" $out/e2e_t$t.txt | head -12; tail -1 $out/e2e_t$t.txt
done
