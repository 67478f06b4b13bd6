#!/bin/bash
# usage: scripts/exp_bench.sh variant1 variant2 ...   (libs under blutils_amd/lib/exp/lib_<variant>.so; "base" = the product build)
# Timing-only A/B of kernel experiments on the C3 workload; parity gate off for builds that skip work.
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so; fi
  python bench.py --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline $EXP_BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v', 'kernel_ms=%.3f'%r['kernel_ms'], 'GB/s=%.0f'%r['achieved'], 'Mq/s=%.0f'%d['value'])"
done
