#!/usr/bin/env python3
"""Stream-kernel time on the C3 table as generated (top rows anywhere in the segment) and with every query's hits sorted
by bit-score, best first — the order BLAST writes them in."""
import sys
sys.path.insert(0, '.')
import torch
from blutils_amd import engine, synth
from tests import helpers as H
tax = synth.make_taxonomy(2400000, synth.SEEDS["C3"])
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
hits = synth.make_hits(tax, 10000000, synth.SEEDS["C3"], 50, device="cuda", columns="milli")
hits.tax_row = t.engine_rows(hits.tax_row).contiguous()
out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device="cuda")
def timeit(d):
    for _ in range(2): engine.run_consensus_device(t, d, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): engine.run_consensus_device(t, d, out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5
print("as generated      : %.3f ms" % timeit(hits.as_dict("packed", tax=t)))
order = torch.argsort(hits.bitscore.view(-1, 50), dim=1, descending=True, stable=True)
order = (order + torch.arange(hits.n_queries, device="cuda").view(-1, 1) * 50).reshape(-1)
for name in ("bitscore", "tax_row", "pident_milli", "align_len", "acc_rank"):
    setattr(hits, name, getattr(hits, name)[order].contiguous())
print("sorted, best first: %.3f ms" % timeit(hits.as_dict("packed", tax=t)))
print("sorted, columns   : %.3f ms" % timeit(hits.as_dict("milli")))
