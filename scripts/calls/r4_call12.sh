#!/bin/bash
# round 4, call 12: final profiles, part 1 — kernel trace + PMC passes of C3 (packed) and C5, kernel trace of the zymo-like table
cd "$GRAFT_REPO_ROOT" || exit 1
scripts/final_profiles.sh r13 "c3:C3-packed:" "c5:C5-packed:--config C5" "zymo:-:--top-group zymo"
ls gpurun_out | grep profiles_r13
