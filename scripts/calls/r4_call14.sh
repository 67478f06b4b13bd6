#!/bin/bash
# round 4, call 14: cache-policy experiments — last load of the reference row non-temporal, offsets non-temporal (variant "head" = this source, no flags)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call14; mkdir -p $out
REPS=7 scripts/ab.sh head lastnt segnt bothnt > $out/ab_c3.txt 2>&1; echo "[c3]"; cat $out/ab_c3.txt
AB_ARGS="--top-group zymo" REPS=3 scripts/ab.sh head lastnt bothnt > $out/ab_zymo.txt 2>&1; echo "[zymo]"; cat $out/ab_zymo.txt
scripts/pmc_variants.sh head lastnt bothnt > $out/pmcv.txt 2>&1; echo "[requests]"; tail -6 $out/pmcv.txt
