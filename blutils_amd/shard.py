"""Multi-GPU: queries shard embarrassingly.  One process per GPU; contiguous query ranges balanced
by HIT count (so skewed segment lengths carry equal bytes); the taxonomy is replicated; every rank runs the same
kernels on its slice; records are concatenated on the host in query order.  No data-path collective: the only
communication is the gather of the 32-byte records to rank 0 (and the barrier/max of the benchmark clock)."""
from __future__ import annotations

from typing import Callable, List, Tuple

import numpy as np


def balanced_query_ranges(seg_off: np.ndarray, n_parts: int) -> List[Tuple[int, int]]:
    """Contiguous [q0, q1) ranges whose hit counts are as equal as the segment boundaries allow."""
    seg = np.asarray(seg_off).astype(np.int64)
    nq = len(seg) - 1
    total = int(seg[-1]) if nq >= 0 else 0
    cuts = [0]
    for p in range(1, n_parts):
        target = total * p // n_parts
        q = int(np.searchsorted(seg, target, side="left"))
        # the boundary nearest to the target, never moving backwards
        if q > 0 and q <= nq and abs(int(seg[q - 1]) - target) <= abs(int(seg[min(q, nq)]) - target):
            q -= 1
        cuts.append(min(max(q, cuts[-1]), nq))
    cuts.append(nq)
    return [(cuts[i], cuts[i + 1]) for i in range(n_parts)]


def slice_table(hits: dict, q0: int, q1: int) -> dict:
    """The rows of queries [q0, q1) with offsets rebased to 0 (numpy arrays or torch tensors)."""
    seg = hits["seg_off"]
    r0, r1 = int(seg[q0]), int(seg[q1])
    # (the packed layout holds four words per hit row: {tax_row, pident_milli, align_len, acc_rank})
    out = {k: (v[4 * r0:4 * r1] if k == "packed" else v[r0:r1]) for k, v in hits.items() if k != "seg_off"}
    out["seg_off"] = seg[q0:q1 + 1] - seg[q0]
    return out


def rebase_records(records: np.ndarray, row0: int) -> np.ndarray:
    """ref_row of a slice's records -> row index in the whole table (records with a reference row only)."""
    out = records.copy()
    has_row = out["ref_row"] != 0xFFFFFFFF
    out["ref_row"][has_row] = (out["ref_row"][has_row].astype(np.int64) + row0).astype(np.uint32)
    return out


def run_sharded(hits: dict, runner: Callable[[dict], np.ndarray], rank: int, world: int, gather: bool = True):
    """Every rank calls this with the same host table; `runner` maps a slice to its records (on the GPU engine:
    engine.run_consensus_host bound to this rank's device).  Rank 0 gets the concatenated records, others None."""
    import torch
    import torch.distributed as dist

    ranges = balanced_query_ranges(hits["seg_off"], world)
    q0, q1 = ranges[rank]
    local = rebase_records(runner(slice_table(hits, q0, q1)), int(hits["seg_off"][q0]))
    if world == 1 or not gather:
        return local
    payload = torch.from_numpy(local.view(np.uint8).copy())
    sizes = [32 * (b - a) for a, b in ranges]
    if dist.get_backend() == "nccl":
        payload = payload.cuda()
        bufs = [torch.empty(s, dtype=torch.uint8, device="cuda") for s in sizes]
    else:
        bufs = [torch.empty(s, dtype=torch.uint8) for s in sizes]
    dist.all_gather(bufs, payload) if len(set(sizes)) == 1 else _all_gather_ragged(bufs, payload, rank, world)
    if rank != 0:
        return None
    return np.concatenate([b.cpu().numpy() for b in bufs]).view(local.dtype)


def _all_gather_ragged(bufs, payload, rank, world):
    import torch.distributed as dist
    for src in range(world):
        if src == rank:
            bufs[src].copy_(payload)
        dist.broadcast(bufs[src], src=src)
