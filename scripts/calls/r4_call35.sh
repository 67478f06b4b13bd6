#!/bin/bash
# round 4, call 35: does any cache policy make the L2 fetch less than a 128-byte line? (scripts/probe/sector_probe.hip); end to end once more
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call35; mkdir -p $out /tmp/blu_probe
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 -o /tmp/blu_probe/sector_probe scripts/probe/sector_probe.hip || exit 1
timeout -k 10 120 /tmp/blu_probe/sector_probe 23 > $out/sector.txt 2>&1; echo "[sector probe] rc=$?"; cat $out/sector.txt
timeout -k 10 120 /tmp/blu_probe/sector_probe 25 > $out/sector25.txt 2>&1; echo "[sector probe, 4 GB] rc=$?"; cat $out/sector25.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > $out/tests.txt 2>&1; rc=$?; echo "[tests] rc=$rc"; tail -3 $out/tests.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|engine|free the" $out/e2e.txt; tail -1 $out/e2e.txt
