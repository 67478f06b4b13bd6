#!/bin/bash
# usage (GPU box, repo root): scripts/ingest_profile.sh r01  -> gpurun_out/ingest_<tag>/: kernel trace of one GPU ingest of a 20 M-row table
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
tag=${1:-r01}
out="$GRAFT_REPO_ROOT/gpurun_out/ingest_$tag"
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" || exit 1
python3 scripts/ingest_bench.py --queries 400000 --threads 16 > $out/cpu.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/ingest_once.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from blutils_amd import pipeline
d = "/tmp/blu_ingest_bench"
print(pipeline.ingest_only(d + "/blast.clustered.tsv", d + "/tax.blucache", False, 0), pipeline.last_ingest_path())
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/ingest_once.py > $out/run.txt 2> $out/run.log
echo "trace rc=$?"
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs cat > $out/kernel_stats.csv
cat $out/run.txt; cut -c1-160 $out/kernel_stats.csv
