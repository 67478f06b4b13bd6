#!/bin/bash
# usage: [REPS=3] scripts/exp_bench.sh variant1 variant2 ...   (libs under blutils_amd/lib/exp/lib_<variant>.so; "base" = product build)
# Timing-only A/B of kernel experiments on the C3 workload, variants interleaved REPS times (each process gets its
# own physical pages: run-to-run spread on one box is a few %, so compare medians).  Parity gate off.
REPS=${REPS:-3}
for r in $(seq $REPS); do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so; fi
  python bench.py --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary $EXP_BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', '%.3f'%r['kernel_ms'])"
done
done | python -c "
import sys, collections, statistics
d=collections.OrderedDict()
for l in sys.stdin:
    k,v=l.split(); d.setdefault(k,[]).append(float(v))
for k,v in d.items(): print('%-10s median %.3f  min %.3f  max %.3f  (%s)'%(k, statistics.median(v), min(v), max(v), ' '.join('%.3f'%x for x in v)))"
