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
