#!/bin/bash
# usage (GPU box): scripts/gpu_call.sh <tag> step...   — runs the named steps in order, each under its own timeout, and
# stops at the first one that was killed by its timeout (no further GPU step after a hang).  Logs under gpurun_out/<tag>/.
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
run() {   # run <name> <timeout s> <command...>
  local name=$1 lim=$2; shift 2
  echo "[gpu_call] $name ..."
  timeout -k 10 "$lim" "$@" > "$out/$name.out" 2> "$out/$name.err"
  local rc=$?
  echo "[gpu_call] $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "[gpu_call] $name was killed at its limit: stopping"; exit 1; fi
  return 0
}
for step in "$@"; do
  case "$step" in
    tests)      run tests 1100 python -m pytest tests -m gpu -x -q ;;
    tests_fast) run tests_fast 600 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py ;;
    bench)      run bench 300 python bench.py ;;
    smoke)      run smoke 300 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench_q)    run bench_q 200 python bench.py --no-cpu-baseline --cpu-sample 100000 ;;
    dist)       BLU_BENCH_FORCE_DIST=1 run dist 300 python bench.py --steps 10 --no-cpu-baseline --cpu-sample 100000 ;;
    zymo)       run zymo 200 python bench.py --top-group zymo --no-cpu-baseline --cpu-sample 100000 ;;
    all50)      run all50 200 python bench.py --top-group all --queries 2000000 --no-cpu-baseline --cpu-sample 100000 ;;
    c5)         run c5 300 python bench.py --config C5 --no-cpu-baseline --cpu-sample 100000 ;;
    c2)         run c2 200 python bench.py --config C2 --no-cpu-baseline --cpu-sample 100000 ;;
    f64)        run f64 300 python bench.py --pident f64 --no-cpu-baseline --cpu-sample 100000 ;;
    h10)        run h10 200 python bench.py --hits-per-query 10 --queries 20000000 --no-cpu-baseline --cpu-sample 100000 ;;
    h100)       run h100 200 python bench.py --hits-per-query 100 --queries 5000000 --no-cpu-baseline --cpu-sample 100000 ;;
    prof_c3)    run prof_c3 900 scripts/profile_round.sh ${tag}_c3 ;;
    prof_c5)    run prof_c5 900 scripts/profile_round.sh ${tag}_c5 --config C5 ;;
    prof_f64)   run prof_f64 900 scripts/profile_round.sh ${tag}_f64 --pident f64 ;;
    exp:*)      v=${step#exp:}; REPS=${REPS:-2} run "exp_${v//,/_}" 600 scripts/exp_bench.sh ${v//,/ } ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
for f in "$out"/*.out; do echo "== $f"; tail -c 1500 "$f"; echo; done
