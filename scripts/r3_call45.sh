#!/bin/bash
# round-3 GPU call 45: GPU suite on the node-id read-back in the kernel without the ring; worklist kernel at 7 waves per SIMD (72 VGPRs, no scratch) against 8
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > gpurun_out/c45_tests.log 2>&1 || { tail -40 gpurun_out/c45_tests.log; exit 1; }
tail -3 gpurun_out/c45_tests.log
REPS=4 AB_ARGS="--config C5" scripts/ab.sh base b7 > gpurun_out/c45_c5.log 2>&1; cat gpurun_out/c45_c5.log
