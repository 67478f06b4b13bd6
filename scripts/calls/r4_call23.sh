#!/bin/bash
# round 4, call 23: what the parse kernel's time is made of — timing-only builds (a BLU_PARSE_X macro that existed only for this call: no numbers / no hashes / no taxid join / no stores / staging only) under the kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call23; mkdir -p $out
python3 scripts/e2e_bench.py --queries 1000000 --reps 1 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; tail -1 $out/e2e.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/ingest_once.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from blutils_amd import pipeline
d = "/tmp/blu_e2e"
try:
    print(pipeline.ingest_only(d + "/blast.1000000x50.clustered.tsv", d + "/tax.blucache", False, 0), pipeline.last_ingest_path())
except Exception as e:
    print("ingest raised", e)
PY
for x in 0 1 2 4 8 16 32 3 63; do
  BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_px$x.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_x$x -- python3 /tmp/ingest_once.py > $out/run_x$x.txt 2> $out/run_x$x.log || { echo "x$x failed"; tail -3 $out/run_x$x.log; }
  f=$(find $out/trace_x$x -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/kernel_stats_x$x.csv
  python3 -c "
import csv, sys
for r in csv.DictReader(open('$out/kernel_stats_x$x.csv')):
    if 'parse_rows' in r['Name']: print('x$x: parse_rows %.3f ms' % (int(r['TotalDurationNs']) / 1e6))
"
  rm -rf $out/trace_x$x
done
