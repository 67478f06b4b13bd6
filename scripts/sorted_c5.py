#!/usr/bin/env python3
"""C5 (Zipf 1..5000 hits per query) as generated — top rows anywhere in a segment — and with every query's hits sorted
by bit-score, best first, the order BLAST writes them in: both kernels' time per run (packed layout)."""
import sys
sys.path.insert(0, '.')
import torch
from blutils_amd import engine, synth
cfg = synth.CONFIGS["C5"]
tax = synth.make_taxonomy(cfg["n_taxa"], synth.SEEDS["C5"], deep=True)
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=0)
hits = synth.make_hits(tax, cfg["n_queries"], synth.SEEDS["C5"], None, zipf=cfg["zipf"], device="cuda", columns="milli")
hits.tax_row = t.engine_rows(hits.tax_row).contiguous()
out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device="cuda")
def timeit(d):
    for _ in range(2): engine.run_consensus_device(t, d, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): engine.run_consensus_device(t, d, out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 10
print(f"as generated: {timeit(hits.as_dict('packed')):.3f} ms ({hits.n_queries} queries, {hits.n_hits} rows)")
seg = hits.seg_off
qid = torch.repeat_interleave(torch.arange(hits.n_queries, device="cuda"), (seg[1:] - seg[:-1]))
key = qid * (1 << 32) + ((1 << 31) - 1 - hits.bitscore.to(torch.int64))
order = torch.argsort(key, stable=True)
del key, qid
for name in ("bitscore", "tax_row", "pident_milli", "align_len", "acc_rank"):
    setattr(hits, name, getattr(hits, name)[order].contiguous())
print(f"best first:   {timeit(hits.as_dict('packed')):.3f} ms")
