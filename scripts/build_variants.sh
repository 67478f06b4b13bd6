#!/bin/bash
# usage: scripts/build_variants.sh "name:-DFLAG ..." ...  -> blutils_amd/lib/exp/lib_<name>.so (experimental builds, same ABI)
set -e
cd "$(dirname "$0")/../blutils_amd/csrc"
mkdir -p ../lib/exp
for v in "$@"; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include --offload-arch=gfx950 \
     -x hip consensus_kernel.hip ingest_gpu.hip pack_kernel.hip taxonomy.cpp api.cpp pipeline.cpp -shared -o ../lib/exp/lib_$n.so $f &
done
wait
ls ../lib/exp
