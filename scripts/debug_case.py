import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blutils_amd import engine, synth
from tests import helpers as H
tax = synth.make_taxonomy(30000, synth.SEEDS["C5"], deep=True)
h = synth.make_hits(tax, 20000, synth.SEEDS["C5"], None, zipf=(1.1, 1, 5000)).numpy()
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="fungi", device=0, taxid=tax.taxid)
seg = h["seg_off"]
got = engine.run_consensus_host(t, seg, h["bitscore"], t.engine_rows(h["tax_row"]), h["pident"], h["align_len"], h["acc_rank"], strategy="relaxed")
exp = H.columnar(tax, h, "fungi", "relaxed", threads=8)
bad = np.unique(np.nonzero(got.view(np.uint8).reshape(-1, 32) != exp.view(np.uint8).reshape(-1, 32))[0])
print("differ:", len(bad), bad)
lens = np.diff(seg)
for q in bad[:4]:
    tk = q // 64
    a, b = int(seg[q]), int(seg[q + 1])
    bs = h["bitscore"][a:b]
    print(" q", q, "task", tk, "lane", q % 64, "len", b - a, "rel row", a - int(seg[tk * 64]), "top rows", np.nonzero(bs == bs.max())[0][:12], "tax", h["tax_row"][a:b][bs == bs.max()][:12])
    print("   got", got[q]); print("   exp", exp[q])
    print("   task lens", lens[tk * 64:(tk + 1) * 64].tolist())
    print("   task seg rel", (seg[tk * 64:(tk + 1) * 64 + 1] - seg[tk * 64]).tolist())
