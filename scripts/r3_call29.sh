#!/bin/bash
# round-3 GPU call 29: the same stamps with the table / kernel kind crossed: C3 on the kernel without the ring, C5 on the one with it
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so
{
echo "== C3 (2 M queries), kernel without the ring"
BLU_STREAM_KIND=noring timeout -k 10 300 python3 scripts/stamps.py --config C3 --queries 2000000 2>&1 | grep -v amdgpu.ids || exit 1
echo "== C3 (2 M queries), kernel with the ring"
BLU_STREAM_KIND=ring timeout -k 10 300 python3 scripts/stamps.py --config C3 --queries 2000000 2>&1 | grep -v amdgpu.ids || exit 1
echo "== C5, kernel with the ring"
BLU_STREAM_KIND=ring timeout -k 10 300 python3 scripts/stamps.py --config C5 2>&1 | grep -v amdgpu.ids || exit 1
} > gpurun_out/c29_stamps.log 2>&1
cat gpurun_out/c29_stamps.log
