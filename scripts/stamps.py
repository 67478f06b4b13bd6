#!/usr/bin/env python3
"""Phase breakdown of the stream kernel from an -DBLU_X_STAMPS experiment build (BLU_CONSENSUS_LIB=.../lib_stamps.so):
runs the C3 table twice and prints the per-wave average of the s_memtime sums (shader cycles) per phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from blutils_amd import engine, synth
CUSTOM = {"domain": 50, "kingdom": 60, "phylum": 75, "class": 80, "order": 85, "family": 92, "genus": 97, "species": 99}
cfg = synth.CONFIGS["C3"]; seed = synth.SEEDS["C3"]
tax = synth.make_taxonomy(cfg["n_taxa"], seed)
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=CUSTOM, device=0)
hits = synth.make_hits(tax, cfg["n_queries"], seed, 50, device="cuda", columns="milli")
hits.tax_row = t.engine_rows(hits.tax_row).contiguous()
hd = hits.as_dict("packed")
out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device="cuda")
for _ in range(2):
    engine.run_consensus_device(t, hd, out)
torch.cuda.synchronize()
name, grid, block = engine.last_launch()
nw = grid * block // 64
st = out[: nw * 64].cpu().numpy().view(np.uint32).reshape(nw, 16)[:, :12].astype(np.float64)
names = ["setup", "phase1", "gather", "phase2a", "ref rows", "run lengths", "codes", "levels+record", "drain ahead", "stores", "p1: ring wait", "p1: list write"]
tasks = hits.n_queries / 64 / nw
tot = st.sum(axis=1).mean()
print(f"{nw} waves, {tasks:.1f} tasks per wave, {tot:.0f} cycles per wave, {tot / tasks:.0f} per task")
for i in range(12):
    print(f"  {names[i]:14s} {st[:, i].mean() / tasks:8.0f} cycles/task  {100 * st[:, i].mean() / tot:5.1f} %")
