#!/bin/bash
# round 4, call 30: worklist kernel at 7 and 6 waves per SIMD (72 / 80 VGPRs, no scratch in any build) against the 8-wave product build, on C5
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_call30; mkdir -p $out
AB_ARGS="--config C5" REPS=5 scripts/ab.sh base b7 b6 > $out/ab_c5.txt 2>&1; echo "[c5]"; cat $out/ab_c5.txt
AB_ARGS="--config C5 --strategy cautious" REPS=3 scripts/ab.sh base b7 b6 > $out/ab_c5c.txt 2>&1; echo "[c5 cautious]"; cat $out/ab_c5c.txt
