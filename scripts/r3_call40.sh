#!/bin/bash
# round-3 GPU call 40: the N > 1 logic of bench.py at HEAD, 2 and 4 ranks sharing the box's one GPU over gloo (C3 and C5 cut into hit-balanced ranges)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
out=gpurun_out/r3c40; mkdir -p $out
export BLU_BENCH_SHARE_GPU=1
: > $out/rehearsal.jsonl
for spec in "2:" "4:" "2:--config C5"; do
  n=${spec%%:*}; args=${spec#*:}
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 100000 --no-secondary $args > $out/ranks.json 2> $out/ranks.err || { tail -5 $out/ranks.err; exit 1; }
  tail -1 $out/ranks.json >> $out/rehearsal.jsonl
  tail -1 $out/ranks.json | cut -c1-400
done
