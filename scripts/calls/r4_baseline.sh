#!/bin/bash
# round 4, call 1: same-box baseline — product build vs "every group narrow" (timing only) vs the build without the ring, + stream ceiling
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r4_baseline; mkdir -p $out
python3 scripts/probe/probe.py > $out/ceiling.txt 2>&1
echo "[ceiling]"; tail -1 $out/ceiling.txt
REPS=3 scripts/ab.sh base nowide > $out/ab_c3.txt 2>&1; cat $out/ab_c3.txt
BLU_STREAM_KIND=noring REPS=2 scripts/ab.sh base > $out/ab_c3_noring.txt 2>&1; echo "[noring forced]"; cat $out/ab_c3_noring.txt
AB_ARGS="--queries 1250000" REPS=3 scripts/ab.sh base nowide > $out/ab_c4.txt 2>&1; echo "[c4 slice]"; cat $out/ab_c4.txt
