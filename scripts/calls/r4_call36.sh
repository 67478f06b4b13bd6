#!/bin/bash
# round 4, call 36: does any cache policy make the L2 fetch less than a 128-byte line? (scripts/probe/sector_probe.hip, buffer-load builtins:
# bounds-checked by the descriptor; the inline-asm version of call 35 faulted and never ran)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call36; mkdir -p $out /tmp/blu_probe
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/blu_probe/sector_probe scripts/probe/sector_probe.hip || exit 1
timeout -k 10 120 /tmp/blu_probe/sector_probe 23 > $out/sector23.txt 2>&1; rc=$?; echo "[sector probe, 1 GB] rc=$rc"; cat $out/sector23.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 /tmp/blu_probe/sector_probe 24 > $out/sector24.txt 2>&1; echo "[sector probe, 2 GB] rc=$?"; cat $out/sector24.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
