#!/bin/bash
# round 3, first GPU call: baselines at HEAD (C3, C4 slices, secondary workloads), ref-row nt A/B, phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c1; mkdir -p $out
(REPS=3 scripts/ab.sh base refnt) > $out/ab_refnt.txt 2>&1
for q in 1250000 625000; do
  AB_ARGS="--queries $q" REPS=2 scripts/ab.sh base > $out/slice_$q.txt 2>&1
done
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so timeout -k 10 200 python3 scripts/stamps.py > $out/stamps_c3.txt 2>&1
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so timeout -k 10 200 python3 scripts/stamps.py --queries 1250000 > $out/stamps_slice.txt 2>&1
timeout -k 10 300 scripts/exp_pmc.sh $out/pmc base refnt > $out/exp_pmc.txt 2>&1
for w in "--top-group zymo" "--config C5" "--pident f64" "--config C2 --graph" "--config C2"; do
  n=$(echo $w | tr -d ' -'); timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary $w > $out/w_$n.json 2> $out/w_$n.err
done
tail -n 20 $out/*.txt
for f in $out/w_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', 'ms_per_step %.4f kernel_ms %.4f frac %.3f value %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"; done
