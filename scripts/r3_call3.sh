#!/bin/bash
# round 3, GPU call 3: ABI v4 (shape hints, level words, 24-byte records): GPU tests without the full-size ones, then A/B against the round-2 library
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c3; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fullsize.py > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -15 $out/tests.txt
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --cpu-sample 200000 > $out/bench_c3.json 2> $out/bench_c3.err; tail -2 $out/bench_c3.err
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --cpu-sample 200000 --pident packed64 > $out/bench_p64.json 2> $out/bench_p64.err; tail -2 $out/bench_p64.err
(REPS=3 scripts/ab.sh r2 base) > $out/ab.txt 2>&1; cat $out/ab.txt
(REPS=2 AB_ARGS="--config C5" scripts/ab.sh r2 base) > $out/ab_c5.txt 2>&1; cat $out/ab_c5.txt
(REPS=2 AB_ARGS="--top-group zymo" scripts/ab.sh r2 base) > $out/ab_zymo.txt 2>&1; cat $out/ab_zymo.txt
for f in $out/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', 'ms_per_step %.4f kernel_ms %.4f frac %.3f value %.0f' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value']))"; done
