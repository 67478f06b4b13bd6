#!/usr/bin/env python3
"""Stream-kernel time on tables whose segment lengths vary inside every wave task (uniform 1..N hits per query)."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from blutils_amd import engine, synth
from tests import helpers as H
tax = synth.make_taxonomy(2400000, synth.SEEDS["C3"])
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
for hi in (20, 50, 100, 200):
    nq = int(500e6 / ((hi + 1) / 2))
    base = synth.make_hits(tax, nq, 7, hi, device="cuda", columns="milli")
    lens = torch.randint(1, hi + 1, (nq,), device="cuda", dtype=torch.int64)
    seg = torch.zeros(nq + 1, dtype=torch.int64, device="cuda"); seg[1:] = torch.cumsum(lens, 0)
    # keep the first lens[q] rows of every query
    keep = (torch.arange(hi, device="cuda").view(1, -1) < lens.view(-1, 1)).reshape(-1)
    cols = {k: getattr(base, k)[keep].contiguous() for k in ("bitscore", "tax_row", "pident_milli", "align_len", "acc_rank")}
    cols["tax_row"] = t.engine_rows(cols["tax_row"]).contiguous()
    rec = torch.stack([cols["tax_row"], cols["pident_milli"], cols["align_len"], cols["acc_rank"]], dim=1).contiguous().reshape(-1)
    d = {"seg_off": seg, "bitscore": cols["bitscore"], "packed": rec}
    out = torch.zeros(32 * nq, dtype=torch.uint8, device="cuda")
    for _ in range(2): engine.run_consensus_device(t, d, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): engine.run_consensus_device(t, d, out)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print(f"uniform 1..{hi}: {nq} queries, {int(seg[-1])} rows: {ms:.3f} ms = {nq / ms / 1e6:.2f} Gq/s, {int(seg[-1]) / ms / 1e6:.0f} G rows/s")
    del base, cols, rec, d, out
    torch.cuda.empty_cache()
