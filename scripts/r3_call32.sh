#!/bin/bash
# round-3 GPU call 32: per-kernel times of C5 under timing-only builds (kernel trace; the records of those builds are wrong)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
for v in base skipp1 no2a no2c nogather empty; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so; fi
  echo "== $v"
  TRACE_ONLY=1 timeout -k 10 300 scripts/profile_round.sh c32_$v --config C5 --no-parity-gate --no-cpu-baseline 2>&1 | python3 -c "
import sys, json, re
t = sys.stdin.read()
i = t.find('[')
for r in json.loads(t[i:]):
    if 'stream_kernel' in r['Name'] or 'long_kernel' in r['Name']:
        print('  %-60s calls %s avg %.1f us' % (r['Name'][10:70], r['Calls'], float(r['AverageNs']) / 1000))
"
done > gpurun_out/c32_attr.log 2>&1
cat gpurun_out/c32_attr.log
