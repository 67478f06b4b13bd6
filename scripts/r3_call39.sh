#!/bin/bash
# round-3 GPU call 39: the whole GPU suite (full-size tables included), smoke(), and the default bench line with its secondary entries
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/c39_tests.log 2>&1 || { tail -40 gpurun_out/c39_tests.log; exit 1; }
tail -3 gpurun_out/c39_tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c39_smoke.log 2>&1 || { tail -20 gpurun_out/c39_smoke.log; exit 1; }
tail -2 gpurun_out/c39_smoke.log
timeout -k 10 500 python3 bench.py > gpurun_out/c39_bench.json 2> gpurun_out/c39_bench.err || { tail -20 gpurun_out/c39_bench.err; exit 1; }
tail -c 6000 gpurun_out/c39_bench.json
