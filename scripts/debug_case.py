import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blutils_amd import engine, synth
from tests import helpers as H
hits = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tax = synth.make_taxonomy(4000, synth.SEEDS["C2"])
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0, taxid=tax.taxid)
base = synth.make_hits(tax, 64 * 30, 200 + hits, 300).numpy()
rng = np.random.default_rng(hits)
capv = rng.choice([16, 32, 64, 128, 256, 300], 30)
caps = np.repeat(capv, 64)
lens = np.minimum(rng.integers(1, 301, 64 * 30), caps)
seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
take = np.concatenate([np.arange(l) + 300 * i for i, l in enumerate(lens)])
h = {k: (base[k][take] if k != "seg_off" else seg) for k in base}
print("caps per task", capv)
got = engine.run_consensus_host(t, h["seg_off"], h["bitscore"], t.engine_rows(h["tax_row"]), h["pident"], h["align_len"], h["acc_rank"], strategy="relaxed")
exp = H.columnar(tax, h, "custom", "relaxed", H.CUSTOM_16S)
bad = np.unique(np.nonzero(got.view(np.uint8).reshape(-1, 32) != exp.view(np.uint8).reshape(-1, 32))[0])
print("differ:", len(bad), bad)
for q in bad[:8]:
    a, b = int(seg[q]), int(seg[q + 1])
    bs = h["bitscore"][a:b]
    print(" q", q, "task", q // 64, "lane", q % 64, "len", b - a, "rel row", a - int(seg[q // 64 * 64]), "top rows", np.nonzero(bs == bs.max())[0], "tax", h["tax_row"][a:b][bs == bs.max()])
    print("   got", got[q]); print("   exp", exp[q])
