#!/bin/bash
# round 3, GPU call 7: tests again (long-kernel fix), per-kernel times of C5 for the round-2 library and HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
out=$PWD/gpurun_out/r3c7; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -6 $out/tests.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in r2 base; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_$v.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -- python3 bench.py --config C5 --steps 10 --warmup 2 --no-parity-gate --no-cpu-baseline --no-secondary > $out/c5_$v.json 2> $out/c5_$v.log
  f=$(find $out/trace_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -8 "$f" | cut -c1-200
done
