#!/bin/bash
# round 4, call 33: the default bench line with the r15 traffic figures published (their stamp matches the kernel source now)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r15_default_bench_line.json 2> gpurun_out/r15_default_bench.log; echo "[bench] rc=$?"; tail -14 gpurun_out/r15_default_bench.log
