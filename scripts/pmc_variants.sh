#!/bin/bash
# usage (GPU box): scripts/pmc_variants.sh variant...   — read requests / L2 hits of the stream kernel per library variant
# (blutils_amd/lib/exp/lib_<variant>.so, "base" = the product build): attribution of the traffic to a change
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT" || exit 1
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_$v.so; fi
  scripts/pmc.sh gpurun_out/pmc_$v "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" > gpurun_out/pmc_$v.log 2>&1
  python3 -c "
import json; d=json.load(open('gpurun_out/pmc_$v/pmc_summary.json'))
k=[x for n,x in d.items() if 'stream_kernel' in n and x.get('TCC_EA0_RDREQ_sum', 0) > 1e6][0]; print('$v', 'RDREQ %.2f M  HIT %.2f M  MISS %.2f M' % (k['TCC_EA0_RDREQ_sum'] / 1e6, k['TCC_HIT_sum'] / 1e6, k['TCC_MISS_sum'] / 1e6))"
done
