#!/usr/bin/env python3
"""Phase breakdown of the stream kernel from an -DBLU_X_STAMPS experiment build (BLU_CONSENSUS_LIB=.../lib_stamps.so):
runs the C3 table twice and prints the per-wave average of the s_memtime sums (shader cycles) per phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from blutils_amd import engine, synth
CUSTOM = {"domain": 50, "kingdom": 60, "phylum": 75, "class": 80, "order": 85, "family": 92, "genus": 97, "species": 99}
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3"); ap.add_argument("--hits-per-query", type=int, default=0); ap.add_argument("--queries", type=int, default=0)
ap.add_argument("--top-group", default="geo")
ap.add_argument("--deep", type=int, default=-1, help="override the config's lineage depth class (0 / 1)")
ap.add_argument("--taxa", type=int, default=0, help="override the number of taxids")
a = ap.parse_args()
cfg = dict(synth.CONFIGS[a.config]); seed = synth.SEEDS[a.config]
if a.hits_per_query: cfg["hits_per_query"] = a.hits_per_query
if a.queries: cfg["n_queries"] = a.queries
if a.deep >= 0: cfg["deep"] = bool(a.deep)
if a.taxa: cfg["n_taxa"] = a.taxa
tax = synth.make_taxonomy(cfg["n_taxa"], seed, deep=cfg["deep"])
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=CUSTOM, device=0)
hits = synth.make_hits(tax, cfg["n_queries"], seed, cfg["hits_per_query"], zipf=cfg["zipf"], device="cuda", columns="milli", top_group=a.top_group)
hits.tax_row = t.engine_rows(hits.tax_row).contiguous()
hd = hits.as_dict("packed", tax=t)
out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device="cuda")
for _ in range(2):
    engine.run_consensus_device(t, hd, out)
torch.cuda.synchronize()
name, grid, block = engine.last_launch()
nw = grid * block // 64
import ctypes
from blutils_amd import _native
buf = np.zeros(nw * 16, dtype=np.uint32)
rc = _native.lib().blu_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(buf)))
assert rc == 0, rc
st = buf.reshape(nw, 16)[:, :12].astype(np.float64)
names = ["setup", "phase1", "gather", "phase2a", "ref rows", "run lengths", "codes", "levels+record", "drain ahead", "stores", "p1: ring wait", "p1: list write"]
tasks = hits.n_queries / 64 / nw
tot = st.sum(axis=1).mean()
print(f"{nw} waves, {tasks:.1f} tasks per wave, {tot:.0f} cycles per wave, {tot / tasks:.0f} per task")
for i in range(12):
    print(f"  {names[i]:14s} {st[:, i].mean() / tasks:8.0f} cycles/task  {100 * st[:, i].mean() / tot:5.1f} %")
