#!/bin/bash
# round 4, call 32: ingest tests (dict_lookup's 12-byte compare), final profiles part 2 (f64 side records, f64 columns), smoke(), the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4_call32
timeout -k 10 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_c_abi.py -m gpu -x -q > gpurun_out/r4_call32/tests.txt 2>&1; rc=$?; echo "[tests] rc=$rc"; tail -3 gpurun_out/r4_call32/tests.txt
[ $rc -eq 0 ] || exit 1
scripts/final_profiles.sh r15 "p64:C3-packed64:--pident packed64" "f64:C3-f64:--pident f64" > gpurun_out/r15_part2.log 2>&1; tail -3 gpurun_out/r15_part2.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r15_smoke.txt 2>&1; echo "[smoke] rc=$?"; tail -1 gpurun_out/r15_smoke.txt
timeout -k 10 600 python bench.py > gpurun_out/r15_default_bench_line.json 2> gpurun_out/r15_default_bench.log; echo "[bench] rc=$?"; tail -16 gpurun_out/r15_default_bench.log
scripts/pipeline_profile.sh r15 2000000 > gpurun_out/r4_call32/profile.txt 2>&1; echo "[pipeline profile] rc=$?"
