#!/bin/bash
# round 3, GPU call 43: judged profiles of HEAD (kernel trace + PMC passes + published traffic) for C3 packed, C5, C3 packed64
cd "$GRAFT_REPO_ROOT" || exit 1
scripts/final_profiles.sh r09 "c3:C3-packed:" "c5:C5-packed:--config C5" "p64:C3-packed64:--pident packed64" > gpurun_out/r3c43_profiles.txt 2>&1
tail -30 gpurun_out/r3c43_profiles.txt
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_r09.json
mkdir -p gpurun_out/profiles_pub && cp profiles/r09_* gpurun_out/profiles_pub/ 2>/dev/null
timeout -k 10 300 scripts/profile_round.sh r09_zymo --top-group zymo > gpurun_out/r3c43_zymo.txt 2>&1; tail -5 gpurun_out/r3c43_zymo.txt
