#!/bin/bash
# round 4, call 37: the streaming ceiling by kernel shape on ONE box (bench.py's probe shapes next to the sector probe's plain loop)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call37; mkdir -p $out /tmp/blu_probe
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/blu_probe/sector_probe scripts/probe/sector_probe.hip || exit 1
timeout -k 10 120 /tmp/blu_probe/sector_probe 24 2>&1 | head -2 | tee $out/sector24.txt
BLU_BENCH_CEILING_TRACE=1 timeout -k 10 300 python -c "
import bench, torch
for gib in (1.0, 2.0, 4.0):
    print(gib, 'GiB:', bench.stream_read_ceiling(torch, gib))
" 2>&1 | tee $out/ceiling.txt
