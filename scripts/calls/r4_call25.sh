#!/bin/bash
# round 4, call 25: memory of a finished call released by a thread of its own; does one file take parallel writers here?
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call25; mkdir -p $out /tmp/blu_e2e
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py tests/test_cli.py -m gpu -x -q > $out/tests.txt 2>&1; echo "[tests] rc=$?"; tail -3 $out/tests.txt
gcc -O2 -o /tmp/blu_e2e/write_probe scripts/probe/write_probe.c -lpthread && { df -T /tmp | tail -1; /tmp/blu_e2e/write_probe /tmp/blu_e2e/wp.out 841; /tmp/blu_e2e/write_probe /tmp/blu_e2e/wp.out 841; } 2>&1 | tee $out/write_probe.txt
timeout -k 10 300 python scripts/e2e_bench.py --reps 5 --dir /tmp/blu_e2e > $out/e2e.txt 2>&1; echo "[e2e] rc=$?"
grep -E "^rep|start-up|upload|engine|render  |writer  |hand|tear-down" $out/e2e.txt; tail -1 $out/e2e.txt
