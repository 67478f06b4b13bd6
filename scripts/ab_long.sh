# same-box A/B of the worklist kernel: C5 and fixed 600 / 1000 / 3000 hits per query (variants under blutils_amd/lib/exp)
for v in "$@"; do
  if [ "$v" = "base" ]; then unset BLU_CONSENSUS_LIB; else export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so; fi
  for r in 1 2; do python bench.py --config C5 --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v C5', round(d['ms_per_step'],4))"; done
  python bench.py --config C5 --pident f64 --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v C5 f64', round(d['ms_per_step'],4))"
  for h in 600 1000 3000; do q=$((200000000/h)); python bench.py --queries $q --hits-per-query $h --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', $h, 'hits', round(d['ms_per_step'],4))"; done
done
