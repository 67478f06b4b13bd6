#!/bin/bash
# usage: scripts/resusage.sh [extra hipcc flags]   — VGPRs / scratch / LDS per kernel of consensus_kernel.hip (compile only, no GPU)
cd "$(dirname "$0")/../blutils_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include --offload-arch=gfx950 -x hip -c consensus_kernel.hip -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import re, sys
cur = {}
rows = []
for l in sys.stdin:
    m = re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    n = r['name']
    m = re.search(r'kernelILi(\d)ELi(\d)E(?:Lb(\d)E)?', n)
    short = ('stream' if 'stream' in n else 'long' if 'long' in n else n[:30])
    if m: short += ' strat=%s layout=%s%s' % (m.group(1), m.group(2), '' if m.group(3) is None else (' ring' if m.group(3) == '1' else ' noring'))
    print('%-40s VGPR %4s  AGPR %3s  scratch %5s  LDS %7s  SGPR %4s  occ %s' % (short, r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]'), r.get('SGPRs'), r.get('Occupancy [waves/SIMD]')))
"
