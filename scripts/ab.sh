#!/bin/bash
# usage (GPU box): [REPS=3] [AB_ARGS="--config C5"] scripts/ab.sh variant...   — kernel time of bench.py per variant, interleaved REPS times on this box
# ("base" = product build, others = blutils_amd/lib/exp/lib_<variant>.so); prints the median per variant.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT" || exit 1
REPS=${REPS:-3}
for r in $(seq $REPS); do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary $AB_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.4f' % d['roofline']['kernel_ms'])"
done
done | python3 -c "
import sys, collections, statistics
d=collections.OrderedDict()
for l in sys.stdin:
    k,v=l.split(); d.setdefault(k,[]).append(float(v))
for k,v in d.items(): print('%-10s median %.4f ms  (%s)'%(k, statistics.median(v), ' '.join('%.4f'%x for x in v)))"
