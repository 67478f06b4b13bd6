# kernel time per hits-per-query setting (the second table of DESIGN.md §4.3), default layout, one box
for cfg in "10 20000000" "20 20000000" "30 15000000" "50 10000000" "100 5000000" "200 2500000" "500 1000000" "600 800000" "1000 500000" "3000 160000"; do
  set -- $cfg
  python bench.py --queries $2 --hits-per-query $1 --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 hits', '$2 q', round(d['ms_per_step'],3), 'ms', round(d['value']/1000,2), 'Gq/s')"
done
python bench.py --config C5 --no-parity-gate --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C5', round(d['ms_per_step'],3), 'ms', round(d['value']/1000,2), 'Gq/s')"
