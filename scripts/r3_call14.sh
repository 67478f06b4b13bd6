#!/bin/bash
# round 3, GPU call 14: does the flat pass cost the mixed tables it is not used on?  prev (no flat code) / base / flat0 (compiled out)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c14; mkdir -p $out
for rep in 1 2; do
for v in prev base flat0; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$GRAFT_REPO_ROOT/blutils_amd/lib/exp/lib_$v.so; fi
  echo "== $v"; python3 scripts/mixed_bench.py 2>/dev/null | tail -2
done
done > $out/mixed.txt 2>&1
cat $out/mixed.txt
(REPS=3 AB_ARGS="--config C5" scripts/ab.sh prev base flat0) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
