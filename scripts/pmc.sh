#!/bin/bash
# usage: scripts/pmc.sh <outdir> "<counters pass 1>" "<counters pass 2>" ...
# one rocprofv3 --pmc run per counter group (gfx950: FETCH_SIZE and WRITE_SIZE do not fit one pass); PMC_BENCH_ARGS = extra
# bench.py arguments (workload / layout)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
out=$1; shift
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out" ;; esac
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-include-regex "blu_" --output-format csv -d "$out/pass$i" -- \
     python3 bench.py --steps 3 --warmup 1 --no-parity-gate --no-cpu-baseline --no-secondary $PMC_BENCH_ARGS > "$out/pass$i.json" 2> "$out/pass$i.log"
  echo "pass $i ($grp) rc=$?"
done
python3 scripts/pmc_summary.py "$out"
