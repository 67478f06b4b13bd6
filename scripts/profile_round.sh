#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh r01
# Produces gpurun_out/profiles_<tag>/: rocprofv3 kernel-trace stats of the default bench command and the PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes) for the traffic figure.
tag=${1:-r01}
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py > $out/bench_under_trace.json 2> $out/bench_under_trace.log
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
scripts/pmc.sh $out/pmc "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" > $out/pmc.log 2>&1
tail -3 $out/pmc.log
python3 scripts/probe/probe.py 12 > $out/stream_read_ceiling.txt 2>&1; tail -1 $out/stream_read_ceiling.txt
