#!/bin/bash
# round 4, call 20: where the text upload's time goes (scripts/probe/upload_probe.hip) on a 6.7 GB table in the page cache
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call20; mkdir -p $out /tmp/blu_e2e
nproc > $out/nproc.txt; lscpu | grep -E "Model name|Socket|NUMA|^CPU\(s\)" >> $out/nproc.txt; free -g >> $out/nproc.txt; cat $out/nproc.txt
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/blu_e2e/upload_probe scripts/probe/upload_probe.hip -lpthread || exit 1
gcc -O2 -o /tmp/blu_e2e/gen_blast scripts/tools/gen_blast.c || exit 1
f=/tmp/blu_e2e/blast.2000000x50.clustered.tsv
/tmp/blu_e2e/gen_blast table $f 2000000 50 300000 1 clustered || exit 1
ls -la $f
for cfg in "1 8" "2 8" "3 8" "3 4" "2 16" "6 8"; do
  set -- $cfg
  echo "== threads $1, piece $2 MiB" | tee -a $out/probe.txt
  timeout -k 10 200 /tmp/blu_e2e/upload_probe $f $1 $2 >> $out/probe.txt 2>&1 || { echo "probe failed"; tail -5 $out/probe.txt; exit 1; }
done
cat $out/probe.txt
echo "[ingest tests]"
timeout -k 10 400 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py -m gpu -x -q > $out/tests.txt 2>&1; echo "rc=$?"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 3 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"; grep -E "^rep|upload|parse|start-up" $out/e2e.txt; tail -1 $out/e2e.txt
