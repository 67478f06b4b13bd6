#!/bin/bash
# round-3 GPU call 27: is the C5 task's round-trip time a matter of the lineage table (row width, size)?
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_stamps.so
for v in "--deep 0" "--deep 1 --taxa 600000" "--deep 0 --taxa 600000"; do
  echo "== C5 $v"
  timeout -k 10 300 python3 scripts/stamps.py --config C5 $v 2>&1 | grep -v amdgpu.ids || exit 1
done > gpurun_out/c27_stamps.log 2>&1
cat gpurun_out/c27_stamps.log
