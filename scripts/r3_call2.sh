#!/bin/bash
# round 3, GPU call 2: what each phase of the stream kernel costs (timing-only builds that skip it) and early codes loads
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c2; mkdir -p $out
(REPS=3 scripts/ab.sh base early1 early2 skip2c skiplev skip2a skipg scanmin) > $out/ab.txt 2>&1
cat $out/ab.txt
