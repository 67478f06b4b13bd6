"""One periodic table of `blocks` copies of a base block, run through the device path; prints whether the records repeat.
Used to find the size at which tests/test_gpu_fullsize.py::test_table_near_the_row_limit_is_periodic first failed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blutils_amd import engine, synth

blocks, kind, layout = int(sys.argv[1]), sys.argv[2], sys.argv[3]
tax = synth.make_taxonomy(30000, synth.SEEDS["C5"], deep=True)
t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=0)
if kind == "zipf":
    dh = synth.make_hits(tax, 100_000, synth.SEEDS["C5"], None, zipf=(1.1, 1, 3000), device="cuda")
elif kind == "zipf400":
    dh = synth.make_hits(tax, 400_000, synth.SEEDS["C5"], None, zipf=(1.1, 1, 400), device="cuda")
else:
    dh = synth.make_hits(tax, 500_000, synth.SEEDS["C5"], 50, device="cuda")
Q0, H0 = dh.n_queries, dh.n_hits
dh.tax_row = t.engine_rows(dh.tax_row.clone()).contiguous()
base = dh.as_dict(layout)
k = blocks if blocks > 0 else ((1 << 32) - 2) // H0
Hn, Q = k * H0, k * Q0
big = {}
for name, col in base.items():
    if name == "seg_off":
        seg = (col[:-1].to(torch.int64)[None, :] + (torch.arange(k, device="cuda", dtype=torch.int64) * H0)[:, None]).reshape(-1)
        big[name] = torch.cat([seg, torch.tensor([Hn], device="cuda", dtype=torch.int64)])
    else:
        big[name] = col.repeat(k)
torch.cuda.synchronize()
print(f"[probe] {kind} {layout}: {k} x ({Q0} q, {H0} rows) = {Q} q, {Hn} rows ({Hn / 2**31:.3f} x 2^31)", flush=True)
out = torch.zeros(32 * Q, dtype=torch.uint8, device="cuda")
engine.run_consensus_device(t, big, out, strategy="relaxed")
torch.cuda.synchronize()
w = out.view(torch.int32).view(k, Q0, 8)
other = torch.ones(8, dtype=torch.bool, device="cuda"); other[3] = False
same = bool((w[:, :, other] == w[0][None, :, other]).all())
print(f"[probe] ok, periodic={same}", flush=True)
