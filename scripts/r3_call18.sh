#!/bin/bash
# round 3, GPU call 18: N > 1 logic of bench.py rehearsed with 4 ranks (the box allows 6 processes on its GPU, the launcher counts) sharing the one GPU over gloo: C3 cut into 6 hit-balanced ranges
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r3c18; mkdir -p $out
export BLU_BENCH_SHARE_GPU=1
for n in 4; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus $n --steps 10 --warmup 3 --no-cpu-baseline --cpu-sample 100000 --no-secondary > $out/ranks$n.json 2> $out/ranks$n.err
tail -2 $out/ranks$n.err | cut -c1-200; tail -1 $out/ranks$n.json | cut -c1-600
done
