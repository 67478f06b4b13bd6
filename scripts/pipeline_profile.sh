#!/bin/bash
# usage (GPU box): scripts/pipeline_profile.sh <tag> [queries]  -> gpurun_out/pipeline_<tag>/kernel_stats.csv: rocprofv3 kernel trace of one
# whole use-case call (GPU ingest + taxonomy + engine + top-score rows; JSONL to a file) on the e2e_bench.py inputs.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
tag=${1:-r05}; nq=${2:-2000000}
out="$GRAFT_REPO_ROOT/gpurun_out/pipeline_$tag"
mkdir -p "$out"
cd "$GRAFT_REPO_ROOT" || exit 1
python3 scripts/e2e_bench.py --queries "$nq" --reps 1 > "$out/e2e.txt" 2>&1     # (also leaves the inputs under /tmp/blu_e2e)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat > /tmp/pipeline_once.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from blutils_amd import pipeline
d = "/tmp/blu_e2e"
_, st = pipeline.build_consensus_identities(d + "/blast.${nq}x50.clustered.tsv", d + "/tax.blucache", "bacteria", "relaxed", out_format="jsonl",
                                            lenient=True, parse=False, out_path=d + "/profiled.jsonl")
print(st, pipeline.last_ingest_path())
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 /tmp/pipeline_once.py > "$out/run.txt" 2> "$out/run.log"
echo "trace rc=$?"
find "$out/trace" -name "*kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
tail -2 "$out/run.txt"; cut -c1-150 "$out/kernel_stats.csv" | head -40
