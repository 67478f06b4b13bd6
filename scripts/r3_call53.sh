#!/bin/bash
# round-3 GPU call 53: at HEAD, for the record — kernel time per hits-per-query setting, then the whole use-case end to end (fresh process per repetition)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 bash scripts/hits_sweep.sh > gpurun_out/c53_sweep.log 2>&1; cat gpurun_out/c53_sweep.log
timeout -k 10 500 python3 scripts/e2e_bench.py --reps 3 > gpurun_out/c53_e2e.log 2>&1; tail -25 gpurun_out/c53_e2e.log
