import numpy as np, sys
sys.path.insert(0, '.')
from blutils_amd import engine, synth
from tests import helpers as H
tax = synth.make_taxonomy(30000, synth.SEEDS["C5"], deep=True)
hits = synth.make_hits(tax, 20000, synth.SEEDS["C5"], None, zipf=(1.1, 1, 5000)).numpy()
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="fungi", device=0)
for strategy in ("relaxed", "cautious"):
    got = engine.run_consensus_host(t, hits["seg_off"], hits["bitscore"], hits["tax_row"], hits["pident"], hits["align_len"], hits["acc_rank"], strategy=strategy)
    exp = H.columnar(tax, hits, "fungi", strategy, threads=8)
    bad = np.nonzero(got.view(np.uint8).reshape(-1,32) != exp.view(np.uint8).reshape(-1,32))[0]
    for q in np.unique(bad)[:6]:
        s, e = int(hits["seg_off"][q]), int(hits["seg_off"][q+1])
        bs = hits["bitscore"][s:e]; M = bs.max(); top = np.nonzero(bs == M)[0]
        print(strategy, "query", q, "n", e - s, "k", len(top), "got", got[q], "exp", exp[q])
        lens = np.diff(tax.lin_off.astype(np.int64))
        for i in top[:40]:
            r = s + i
            print("   pos", i, "row", r, "len", lens[hits["tax_row"][r]], "pid", hits["pident"][r], "aln", hits["align_len"][r], "acc", hits["acc_rank"][r].view(np.uint32) if hasattr(hits["acc_rank"][r],'view') else hits["acc_rank"][r], "tax", hits["tax_row"][r])
