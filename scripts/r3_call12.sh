#!/bin/bash
# round 3, GPU call 12: flat pass, query-aligned single-pass steps: tests, C5 / mixed A/B against the build before the flat pass
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c12; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -5 $out/tests.txt
(REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
python3 scripts/mixed_bench.py > $out/mixed.txt 2>&1; tail -4 $out/mixed.txt
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_prev.so python3 scripts/mixed_bench.py > $out/mixed_prev.txt 2>&1; tail -4 $out/mixed_prev.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/$out/trace -- python3 bench.py --config C5 --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary > $out/c5.json 2> $out/c5.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3c12/trace/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'blu' in r['Name']: print(r['Name'][:70].ljust(72), r['Calls'], r['AverageNs'])
PY
