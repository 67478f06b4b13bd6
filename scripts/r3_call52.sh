#!/bin/bash
# round-3 GPU call 52: HEAD — judged profiles (r12: C3, C5, f64 side records with PMC passes; zymo-like trace), the whole GPU suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
scripts/final_profiles.sh r12 "c3:C3-packed:" "c5:C5-packed:--config C5" "p64:C3-packed64:--pident packed64" > gpurun_out/r3c52_profiles.txt 2>&1 || { tail -20 gpurun_out/r3c52_profiles.txt; exit 1; }
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic_r12.json
mkdir -p gpurun_out/profiles_pub && cp profiles/r12_* gpurun_out/profiles_pub/ 2>/dev/null
TRACE_ONLY=1 timeout -k 10 300 scripts/profile_round.sh r12_zymo --top-group zymo > gpurun_out/r3c52_zymo.txt 2>&1 || { tail -5 gpurun_out/r3c52_zymo.txt; exit 1; }
echo "profiles done"
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/c52_tests.log 2>&1 || { tail -40 gpurun_out/c52_tests.log; exit 1; }
tail -2 gpurun_out/c52_tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/c52_smoke.log 2>&1 || { tail -20 gpurun_out/c52_smoke.log; exit 1; }
tail -1 gpurun_out/c52_smoke.log
timeout -k 10 500 python3 bench.py > gpurun_out/c52_bench.json 2> gpurun_out/c52_bench.err || { tail -20 gpurun_out/c52_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/c52_bench.json").read().strip().splitlines()[-1])
print("C3", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for e in d["secondary"]:
    print(e["workload"][:40], round(e["kernel_ms"], 4), round(e["value"]), round(e["frac"], 3))
PY
