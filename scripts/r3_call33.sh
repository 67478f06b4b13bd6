#!/bin/bash
# round-3 GPU call 33: what an EMPTY C5 stream-kernel task still pays (timing-only builds)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
REPS=1 AB_ARGS="--config C5" scripts/ab.sh base empty e_nopush e_plainst e_nostore e_noflat > gpurun_out/c33_ab.log 2>&1; cat gpurun_out/c33_ab.log
