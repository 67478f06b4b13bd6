#!/bin/bash
# round 3, GPU call 20: staging buffers kept with the handle: tests, host-pointer path rate
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c20; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1; echo "tests rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
timeout -k 10 300 python3 scripts/host_path_bench.py > $out/host_path.txt 2>&1; tail -6 $out/host_path.txt
BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_prev.so timeout -k 10 300 python3 scripts/host_path_bench.py > $out/host_path_prev.txt 2>&1; tail -6 $out/host_path_prev.txt
