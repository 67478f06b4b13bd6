#!/bin/bash
# round 4, call 13: final profiles, part 2 — f64 side records and f64 columns (kernel trace + PMC), then smoke() and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
scripts/final_profiles.sh r13 "p64:C3-packed64:--pident packed64" "f64:C3-f64:--pident f64" > gpurun_out/r13_part2.log 2>&1; tail -3 gpurun_out/r13_part2.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r13_smoke.txt 2>&1; echo "[smoke] rc=$?"; tail -1 gpurun_out/r13_smoke.txt
timeout -k 10 500 python bench.py > gpurun_out/r13_default_bench_line.json 2> gpurun_out/r13_default_bench.log; echo "[bench] rc=$?"; tail -16 gpurun_out/r13_default_bench.log
