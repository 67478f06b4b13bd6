"""N > 1 path on CPU: two gloo ranks shard one table by hit-balanced query ranges, each computes its slice,
records are gathered to rank 0 and must equal the single-process result.  The per-slice compute is the columnar
oracle here (there is no GPU in this container; the engine has no CPU path) — what is under test is the product's
partition / rebase / gather logic in blutils_amd/shard.py, which the GPU run uses unchanged."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blutils_amd import shard, synth
from tests import helpers as H


def test_balanced_ranges_follow_hit_counts():
    seg = np.concatenate([[0], np.cumsum([1, 1, 1, 5000, 1, 1, 2000, 1, 3000, 1, 1])])
    for parts in (1, 2, 3, 4, 8):
        r = shard.balanced_query_ranges(seg, parts)
        assert r[0][0] == 0 and r[-1][1] == len(seg) - 1
        assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    r2 = shard.balanced_query_ranges(seg, 2)
    loads = [int(seg[b] - seg[a]) for a, b in r2]
    assert max(loads) <= 0.6 * seg[-1]
    # empty table / more parts than queries
    assert shard.balanced_query_ranges(np.array([0]), 3) == [(0, 0)] * 3
    assert shard.balanced_query_ranges(np.array([0, 4]), 3)[-1][1] == 1


def _worker(rank, world, port, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tax = synth.make_taxonomy(3000, seed, deep=True)
        hits = synth.make_hits(tax, 700, seed + 1, None, zipf=(1.1, 1, 400), p_unmatched=0.002).numpy()
        runner = lambda sl: H.columnar(tax, sl, "bacteria", "relaxed", threads=1)
        got = shard.run_sharded(hits, runner, rank, world)
        if rank == 0:
            exp = H.columnar(tax, hits, "bacteria", "relaxed", threads=2)
            q.put(bool(got.tobytes() == exp.tobytes()))
        else:
            assert got is None
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 31, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


# ---- bench.py --gpus N: the launcher -------------------------------------------------------------------------------------
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_flag_is_not_decorative():
    """`--gpus N` with no launcher either starts N ranks or fails: it never prints an n_gpus-1 line."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU node runs the real thing (test_bench_self_launch_two_ranks_on_one_card covers one GPU)")
    r = _bench(["--gpus", "2", "--steps", "1"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "GPU" in r.stderr
    # a launcher's world size that disagrees with --gpus is refused before anything is measured
    r = _bench(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_on_one_card():
    """`python bench.py --gpus 2` starts its two ranks itself (fresh children; the parent makes no GPU call).  On a one-GPU box
    the ranks share cuda:0 and meet over gloo (BLU_BENCH_SHARE_GPU=1): what is checked is the launcher, the cut of ONE table
    over the ranks, rank 0's parity gate on its shard and the fields of the line — not a scaling figure."""
    import json
    r = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--queries", "200000", "--taxa", "50000", "--cpu-sample", "20000",
                "--no-cpu-baseline", "--no-secondary"], {"BLU_BENCH_SHARE_GPU": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["queries_total"] == 200000
    assert d["config"]["queries_rank0"] == 100000 and "world_size 2" in d["config"]["process_group"]
