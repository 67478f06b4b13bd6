#!/bin/bash
# usage (GPU box): scripts/final_profiles.sh <tag> <suffix:key:bench args>...
#   e.g. scripts/final_profiles.sh r04 "c3:C3-packed:" "c5:C5-packed:--config C5" "f64:C3-f64:--pident f64"
# Per workload: kernel trace + PMC passes (profile_round.sh), the traffic figure published on the box, then the kernel trace
# again so that the bench line of gpurun_out/profiles_<tag>_<suffix>_t/ carries that figure.  A key of "-" = trace only.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
for spec in "$@"; do
  suffix=${spec%%:*}; rest=${spec#*:}; key=${rest%%:*}; args=${rest#*:}
  echo "== $suffix ($key) args: $args"
  if [ "$key" = "-" ]; then
    TRACE_ONLY=1 timeout -k 10 300 scripts/profile_round.sh "${tag}_${suffix}_t" $args || exit 1
    continue
  fi
  timeout -k 10 700 scripts/profile_round.sh "${tag}_${suffix}" $args || exit 1
  python3 scripts/publish_profiles.py "${tag}_${suffix}" "$key" > /dev/null || exit 1
  TRACE_ONLY=1 timeout -k 10 300 scripts/profile_round.sh "${tag}_${suffix}_t" $args || exit 1
done
