#!/bin/bash
# round 3, GPU call 4: same-box A/B of the round-2 library against HEAD (C3, C4 slice, C5, zymo), cache-policy variants of the reference-row loads, PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c4; mkdir -p $out
(REPS=3 scripts/ab.sh r2 base refsc1 refsc0) > $out/ab.txt 2>&1; cat $out/ab.txt
(REPS=2 AB_ARGS="--queries 1250000" scripts/ab.sh r2 base) > $out/ab_slice.txt 2>&1; cat $out/ab_slice.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh r2 base) > $out/ab_c5.txt 2>&1; cat $out/ab_c5.txt
(REPS=2 AB_ARGS="--top-group zymo" scripts/ab.sh r2 base) > $out/ab_zymo.txt 2>&1; cat $out/ab_zymo.txt
PMC_BENCH_ARGS="--no-secondary" timeout -k 10 500 scripts/pmc.sh $out/pmc "FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" > $out/pmc.txt 2>&1; tail -40 $out/pmc.txt
