#!/bin/bash
# round 3, GPU call 5: what the range-minimum lookups of wide groups cost (timing-only builds), r2 vs HEAD on one box
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c5; mkdir -p $out
(REPS=3 scripts/ab.sh r2 base nowide skiprun) > $out/ab.txt 2>&1; grep median $out/ab.txt
(REPS=2 AB_ARGS="--queries 1250000" scripts/ab.sh r2 base) > $out/ab_slice.txt 2>&1; grep median $out/ab_slice.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh r2 base) > $out/ab_c5.txt 2>&1; grep median $out/ab_c5.txt
(REPS=2 AB_ARGS="--top-group zymo" scripts/ab.sh r2 base) > $out/ab_zymo.txt 2>&1; grep median $out/ab_zymo.txt
export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_nowide.so
PMC_BENCH_ARGS="--no-secondary" timeout -k 10 300 scripts/pmc.sh $out/pmc "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" > $out/pmc.txt 2>&1; grep -A8 "stream_kernel<1, 2, true>" $out/pmc.txt
