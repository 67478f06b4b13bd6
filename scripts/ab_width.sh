# same-box A/B of the streamed-width cost model (BLU_LONG_COST variants under blutils_amd/lib/exp)
for v in prev c96 c128 c192; do
  export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so
  for r in 1 2; do python bench.py --config C5 --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5 $v', round(d['ms_per_step'],4))"; done
  echo "mixed $v"; python scripts/mixed_bench.py 2>/dev/null | cut -d: -f1,3
done
