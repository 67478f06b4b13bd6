#!/bin/bash
# round 3, GPU call 13: why the worklist kernel of C5 takes 298 us now and 271 us with the round-2 library: hinted vs hand-made records
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c13; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in hinted plain r2; do
  unset BLU_CONSENSUS_LIB BLU_BENCH_NO_HINTS
  [ "$v" = "plain" ] && export BLU_BENCH_NO_HINTS=1
  [ "$v" = "r2" ] && export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_r2.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$out/trace_$v -- python3 bench.py --config C5 --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary > $out/c5_$v.json 2> $out/c5_$v.log
  python3 - "$v" <<'PY'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob(f'gpurun_out/r3c13/trace_{v}/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'consensus' in r['Name']: print(v, r['Name'][:60].ljust(62), r['Calls'], r['AverageNs'])
PY
done
python3 scripts/mixed_bench.py > $out/mixed.txt 2>&1; tail -1 $out/mixed.txt
