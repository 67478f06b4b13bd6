import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from blutils_amd import engine, synth
from tests import helpers as H
tax = synth.make_taxonomy(2400000, synth.SEEDS["C3"])
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
hits = synth.make_hits(tax, 2000000, synth.SEEDS["C3"], 50, device="cuda", columns="milli")
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
print("geometric(0.35) top groups: %.3f ms" % timeit(hits.as_dict("packed", tax=t)))
# make the top group larger: the top `g` rows of every query share the top score
bs = hits.bitscore.clone().view(-1, 50)
for g in (5, 10, 15, 20, 25, 30, 40, 50):
    b2 = bs.clone()
    top = b2.max(dim=1, keepdim=True).values
    b2[:, :g] = top
    hits.bitscore = b2.reshape(-1).contiguous()
    ms = timeit(hits.as_dict("packed", tax=t))
    st = engine.records_from_tensor(out)["status"]
    print("top group >= %2d rows: %.3f ms  (%.0f Mq/s)" % (g, ms, hits.n_queries / ms / 1e3))
