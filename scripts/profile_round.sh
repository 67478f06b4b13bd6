#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag> [bench.py arguments of the workload, e.g. --config C5 | --pident f64]
# Produces gpurun_out/profiles_<tag>/: rocprofv3 kernel-trace stats of `bench.py <args>` and the PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) for the traffic figure, plus the sha256 of the kernel
# source they were taken with.  scripts/publish_profiles.py <tag> <key> copies the judged summaries into profiles/.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
tag=${1:-r02}; shift
out="$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
cat blutils_amd/csrc/consensus_kernel.hip blutils_amd/csrc/blu_internal.h | sha256sum | cut -d' ' -f1 > "$out/kernel_sha256.txt"
echo "$*" > "$out/bench_args.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-secondary "$@" > "$out/bench_under_trace.json" 2> "$out/bench_under_trace.log"
echo "trace rc=$?"
python3 - "$out" <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "blu" in r["Name"]]
with open(out + "/kernel_stats_blu.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else ["Name"])
    w.writeheader()
    for r in keep:
        w.writerow(r)
print(json.dumps([{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in keep], indent=1))
PY
[ -n "$TRACE_ONLY" ] && exit 0   # (TRACE_ONLY=1: the bench line and kernel stats again, e.g. after profiles/hbm_traffic.json was published)
PMC_BENCH_ARGS="$*" scripts/pmc.sh "$out/pmc" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" > "$out/pmc.log" 2>&1
tail -3 "$out/pmc.log"
