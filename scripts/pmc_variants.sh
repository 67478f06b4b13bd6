for v in notab noref p1only; do
  export BLU_CONSENSUS_LIB=$PWD/blutils_amd/lib/exp/lib_$v.so
  scripts/pmc.sh gpurun_out/pmc_$v "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum" > gpurun_out/pmc_$v.log 2>&1
  python3 -c "
import json; d=json.load(open('gpurun_out/pmc_$v/pmc_summary.json'))
k=[v for n,v in d.items() if 'stream' in n][0]; print('$v', k['TCC_EA0_RDREQ_sum'])"
done
