#!/bin/bash
# usage (GPU box): scripts/exp_pmc.sh <outdir> variant...   ("base" = product build; others = blutils_amd/lib/exp/lib_<variant>.so)
# Per variant: kernel time of the C3 bench (no profiler) and one PMC pass of SQ instruction / wait counters of the stream kernel.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
out=$1; shift
case "$out" in /*) ;; *) out="$GRAFT_REPO_ROOT/$out" ;; esac
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_$v.so; fi
  timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary $EXP_BENCH_ARGS > "$out/$v.json" 2> "$out/$v.log" || { echo "$v bench failed"; continue; }
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-include-regex "blu_consensus_stream" --output-format csv -d "$out/pmc_$v" -- \
     python3 bench.py --steps 3 --warmup 1 --no-parity-gate --no-cpu-baseline --no-secondary $EXP_BENCH_ARGS > /dev/null 2> "$out/pmc_$v.log"
  python3 - "$out" "$v" <<'PY'
import csv, glob, json, sys, collections
out, v = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/{v}.json").read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(x) / len(x) for k, x in acc.items()}
q = d["config"]["queries_rank0"] / 64
print("%-10s %.3f ms | per task: VALU %.0f SALU %.0f LDS %.0f | wait %.2f active %.2f valu-busy/simd %.2f" % (
    v, d["roofline"]["kernel_ms"], c.get("SQ_INSTS_VALU", 0) / q, c.get("SQ_INSTS_SALU", 0) / q, c.get("SQ_INSTS_LDS", 0) / q,
    c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1),
    c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / (d["roofline"]["kernel_ms"] * 1e-3 * 2.3e9)))
PY
done
