import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from blutils_amd import engine, synth
from tests import helpers as H
cfg = synth.CONFIGS["C3"]; seed = synth.SEEDS["C3"]
Q = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n_queries"]
tax = synth.make_taxonomy(cfg["n_taxa"], seed)
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
dh = synth.make_hits(tax, Q, seed, 50, device="cuda", columns="milli")
dh.tax_row = t.engine_rows(dh.tax_row).contiguous()
hits = dh.as_dict("packed")
Hn = dh.n_hits
def run(h):
    out = torch.zeros(32 * Q, dtype=torch.uint8, device="cuda")
    engine.run_consensus_device(t, h, out, strategy="relaxed")
    torch.cuda.synchronize()
    return engine.records_from_tensor(out)
whole = run(hits)
idx = torch.arange(Hn, device="cuda").view(Q, 50).flip(0).reshape(-1)
rev = {"seg_off": hits["seg_off"], "bitscore": hits["bitscore"][idx].contiguous(), "packed": hits["packed"].view(-1, 4)[idx].contiguous().view(-1)}
print("ptrs: bs %x packed %x ; rev bs %x packed %x" % (hits["bitscore"].data_ptr(), hits["packed"].data_ptr(), rev["bitscore"].data_ptr(), rev["packed"].data_ptr()))
# is the permuted table itself right?
chk = torch.tensor([0, 1157822, 1157823, 1157824, 3000000, 6526532, 6526533, 9999999], device="cuda")
for qq in chk.tolist():
    a = rev["packed"].view(-1, 4)[qq * 50: qq * 50 + 50]
    b = hits["packed"].view(-1, 4)[(Q - 1 - qq) * 50: (Q - 1 - qq) * 50 + 50]
    print("query", qq, "permuted rows equal source rows:", bool((a == b).all()), "zeros:", int((a == 0).all()))
for rep in range(1):
    got = run(rev)
    exp = whole[::-1].copy()
    has = exp["ref_row"] != 0xFFFFFFFF
    qidx = np.nonzero(has)[0]
    exp["ref_row"][has] = (qidx * 50 + exp["ref_row"][has] % 50).astype(np.uint32)
    bad = np.unique(np.nonzero(got.view(np.uint8).reshape(-1, 32) != exp.view(np.uint8).reshape(-1, 32))[0])
    print("rep", rep, "differ", len(bad), bad[:20])
bs = rev["bitscore"].view(Q, 50)
top = (bs == bs.max(dim=1, keepdim=True).values).sum(1).cpu().numpy()
print("first bad row byte addr in rev packed: %x ; last bad: %x" % ((rev["packed"].data_ptr() + int(bad[0]) * 50 * 16) if len(bad) else 0, (rev["packed"].data_ptr() + (int(bad[-1]) + 1) * 50 * 16) if len(bad) else 0))
for q in bad[:0]:
    tk = q // 64
    tops = top[tk * 64:(tk + 1) * 64]
    print("q", q, "task", tk, "lane", q % 64, "k", top[q], "task top rows", tops.sum(), "cum before", tops[: q % 64].sum(), "got", got[q], "exp", exp[q])
    print("   tops per query of the task:", tops.tolist())
