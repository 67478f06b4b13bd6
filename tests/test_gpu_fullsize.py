"""BASELINE.json's full-size configuration (C3: 10 M queries x 50 hits, 2.4 M taxids) on the GPU, checked through
size-independent properties of the path plus an oracle comparison of scattered sub-tables:
  * shard invariance: two halves run separately == the whole run (ref_row rebased) — queries are independent;
  * query-permutation equivariance: reversing the order of whole queries reverses the records;
  * every record is internally consistent (status / flags / masks);
  * 40 scattered windows of 2 500 queries are bit-identical to the columnar oracle, and the first four of them — like three of
    C5's worklist windows — also match the string-faithful oracle field by field (round 4)."""
import numpy as np
import pytest

from blutils_amd import engine, shard, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _give_device_memory_back():
    """These tests hold up to ~110 GB in torch's caching allocator; the library allocates with hipMalloc directly."""
    yield
    import gc
    import torch
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c3_table():
    """C3 as bench.py builds it: taxonomy, engine handle, the generated columns (both pident encodings) and the
    desc rows the oracle reads.  Built once for the layouts below."""
    import torch
    cfg = synth.CONFIGS["C3"]
    seed = synth.SEEDS["C3"]
    tax = synth.make_taxonomy(cfg["n_taxa"], seed)
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
    dh = synth.make_hits(tax, cfg["n_queries"], seed, cfg["hits_per_query"], device="cuda")
    desc_rows = dh.tax_row.clone()
    dh.tax_row = t.engine_rows(desc_rows).contiguous()
    yield tax, t, dh, desc_rows
    del dh, desc_rows, t
    import gc
    gc.collect()
    torch.cuda.empty_cache()


def _assert_window_matches_faithful(tax, t, sub, window_records):
    """The engine's records of one window, rendered into the reference's field values, against the STRING-FAITHFUL oracle
    (oracle/blu_oracle.cpp: per-query row structs, lineage strings parsed per top-group row, the reference's own control
    flow) fed the same window — not only the columnar restatement of it.  `sub`: the window's columns with desc rows in
    tax_row and offsets rebased to 0; window_records: the engine's records with ref_row rebased to the window."""
    from oracle import oracle as orc
    _, run = orc.faithful_on_synthetic(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, sub["seg_off"], sub["bitscore"],
                                       sub["tax_row"], sub["pident"], sub["align_len"], sub["acc_rank"], taxon="custom",
                                       strategy="relaxed", custom=H.CUSTOM_16S, threads=8, want_json=True)
    faithful = run._json
    run.close()
    r = H.Renderer(tax, sub, lambda c: t.rank_name(c), lambda c: t.rank_name(c, serde=True),
                   lambda row, lvl: bool(t.row_cutoffs(row)[1][lvl]))
    assert len(faithful) == len(window_records)
    for q in range(len(window_records)):
        H.assert_matches_faithful(r.render(window_records[q]), faithful[q], q)


def _reverse_queries(hits, n_queries, hits_per_query):
    """The table with the order of whole queries reversed (rows inside a query keep file order); seg_off untouched.
    Done in slices of 2^18 queries: torch's indexing kernels misplace rows of a tensor of more than 2^31 elements (the
    8 GB packed column) on this stack — a property of the harness, found when this test first ran on the packed layout."""
    import torch
    out = {}
    for k, v in hits.items():
        if k == "seg_off":
            out[k] = v
            continue
        width = hits_per_query * (4 if k == "packed" else 1)
        src = v.view(n_queries, width)
        dst = torch.empty_like(src)
        for a in range(0, n_queries, 1 << 18):
            b = min(n_queries, a + (1 << 18))
            dst[n_queries - b:n_queries - a] = src[a:b].flip(0)
        out[k] = dst.view(-1)
    return out


@pytest.mark.parametrize("layout", ["packed", "f64"])
def test_c3_full_size_properties(c3_table, layout):
    """`packed` is the layout bench.py times; `f64` the canonical one of BASELINE.md."""
    import torch
    tax, t, dh, desc_rows = c3_table
    Q, Hn = dh.n_queries, dh.n_hits
    hits = dh.as_dict(layout)

    def run(h, nq):
        out = torch.zeros(32 * nq, dtype=torch.uint8, device="cuda")
        engine.run_consensus_device(t, h, out, strategy="relaxed")
        torch.cuda.synchronize()
        return engine.records_from_tensor(out)

    whole = run(hits, Q)
    # --- internal consistency
    st = whole["status"]
    ok = st <= 1
    assert ok.mean() > 0.98 and (st == 0).sum() > 5_000_000 and (st == 1).sum() > 2_500_000
    assert (whole["ref_row"][ok] < Hn).all()
    assert (whole["ref_row"][ok] // 50 == np.nonzero(ok)[0]).all()            # the reference row belongs to its query
    assert (whole["level_mask"][st == 1] != 0).all()
    assert ((whole["flags"][st == 1] & 1) == 0).all()                          # single match: never "mutated"
    # --- shard invariance
    half = Q // 2
    parts = []
    for q0, q1 in ((0, half), (half, Q)):
        sl = shard.slice_table(hits, q0, q1)
        sl = {k: v.contiguous() for k, v in sl.items()}
        parts.append(shard.rebase_records(run(sl, q1 - q0), q0 * 50))
    assert np.concatenate(parts).tobytes() == whole.tobytes()
    del parts
    # --- permutation equivariance: reverse the order of whole queries (rows inside a query keep file order)
    rev = _reverse_queries(hits, Q, 50)
    assert bool((rev["bitscore"][:50] == hits["bitscore"][-50:]).all()) and bool((rev["bitscore"][-50:] == hits["bitscore"][:50]).all())
    got = run(rev, Q)
    exp = whole[::-1].copy()
    has = exp["ref_row"] != 0xFFFFFFFF
    qidx = np.nonzero(has)[0]
    exp["ref_row"][has] = (qidx * 50 + exp["ref_row"][has] % 50).astype(np.uint32)
    assert got.tobytes() == exp.tobytes()
    del rev, got, exp
    # --- scattered windows against the oracle
    rng = np.random.default_rng(5)
    n_win = 0
    for q0 in rng.integers(0, Q - 2500, 40):
        q0 = int(q0)
        r0, r1 = q0 * 50, (q0 + 2500) * 50
        sub = {"seg_off": np.arange(0, 2500 * 50 + 1, 50, dtype=np.int64),
               "bitscore": dh.bitscore[r0:r1].cpu().numpy(), "tax_row": desc_rows[r0:r1].cpu().numpy(),
               "pident": dh.pident[r0:r1].cpu().numpy(), "align_len": dh.align_len[r0:r1].cpu().numpy(),
               "acc_rank": dh.acc_rank[r0:r1].cpu().numpy()}
        o = H.columnar(tax, sub, "custom", "relaxed", H.CUSTOM_16S, threads=8)
        assert shard.rebase_records(o, r0).tobytes() == whole[q0:q0 + 2500].tobytes()
        n_win += 1
        if n_win <= 4:           # (the first four windows also against the faithful oracle, field by field)
            _assert_window_matches_faithful(tax, t, sub, shard.rebase_records(whole[q0:q0 + 2500].copy(), -r0))


def test_c5_full_size_properties():
    """BASELINE config 5 at full size: 1 M queries with Zipf(1.1) hit counts 1..5000, 2.4 M deep lineages (depth 25-40),
    packed layout — stream kernel (short and long pass) and worklist kernel together.  Internal consistency, shard
    invariance over hit-balanced ranges, and oracle windows chosen so that every one holds worklist queries (> 512 hits)."""
    import torch
    cfg = synth.CONFIGS["C5"]
    seed = synth.SEEDS["C5"]
    tax = synth.make_taxonomy(cfg["n_taxa"], seed, deep=True)
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=H.CUSTOM_16S, device=0)
    dh = synth.make_hits(tax, cfg["n_queries"], seed, None, zipf=cfg["zipf"], device="cuda")
    Q, Hn = dh.n_queries, dh.n_hits
    desc_rows = dh.tax_row.clone()
    dh.tax_row = t.engine_rows(desc_rows).contiguous()
    hits = dh.as_dict("packed")
    seg = dh.seg_off.cpu().numpy().astype(np.int64)
    lens = np.diff(seg)
    assert lens.max() > 4000 and (lens > 512).sum() > 10_000 and t.max_depth >= 25

    def run(h, nq):
        out = torch.zeros(32 * nq, dtype=torch.uint8, device="cuda")
        engine.run_consensus_device(t, h, out, strategy="relaxed")
        torch.cuda.synchronize()
        return engine.records_from_tensor(out)

    whole = run(hits, Q)
    st = whole["status"]
    ok = st <= 1
    assert ok.mean() > 0.9 and (st == 0).sum() > 300_000 and (st == 1).sum() > 200_000
    ref = whole["ref_row"][ok].astype(np.int64)
    qi = np.nonzero(ok)[0]
    assert ((ref >= seg[qi]) & (ref < seg[qi + 1])).all()                      # the reference row belongs to its query
    assert (whole["level_mask"][st == 1] != 0).all()
    assert ((whole["flags"][st == 1] & 1) == 0).all()
    # --- shard invariance over the hit-balanced ranges of the multi-GPU path (3 ranges: uneven query counts)
    parts = []
    for q0, q1 in shard.balanced_query_ranges(seg, 3):
        sl = {k: v.contiguous() for k, v in shard.slice_table(hits, int(q0), int(q1)).items()}
        parts.append(shard.rebase_records(run(sl, int(q1 - q0)), int(seg[q0])))
    assert np.concatenate(parts).tobytes() == whole.tobytes()
    del parts
    # --- oracle windows: 24 windows of 1500 queries, each starting at a worklist query (> 512 rows)
    rng = np.random.default_rng(55)
    long_q = np.nonzero(lens > 512)[0]
    long_q = long_q[long_q < Q - 1500]
    n_work = n_faith = 0
    for q0 in rng.choice(long_q, 24, replace=False):
        q0 = int(q0)
        q1 = q0 + 1500
        r0, r1 = int(seg[q0]), int(seg[q1])
        sub = {"seg_off": seg[q0:q1 + 1] - r0,
               "bitscore": dh.bitscore[r0:r1].cpu().numpy(), "tax_row": desc_rows[r0:r1].cpu().numpy(),
               "pident": dh.pident[r0:r1].cpu().numpy(), "align_len": dh.align_len[r0:r1].cpu().numpy(),
               "acc_rank": dh.acc_rank[r0:r1].cpu().numpy()}
        o = H.columnar(tax, sub, "custom", "relaxed", H.CUSTOM_16S, threads=8)
        assert shard.rebase_records(o, r0).tobytes() == whole[q0:q1].tobytes()
        if n_faith < 3:          # (three windows also against the faithful oracle, field by field)
            _assert_window_matches_faithful(tax, t, sub, shard.rebase_records(whole[q0:q1].copy(), -r0))
            n_faith += 1
        n_work += int((lens[q0:q1] > 512).sum())
    assert n_work >= 24 * 100


@pytest.mark.parametrize("layout", ["packed", "f64"])
def test_table_near_the_row_limit_is_periodic(layout):
    """The ABI's maximum: just under 2^32 - 1 hit rows in one call (include/blu_consensus.h), row offsets far past
    2^31 and byte offsets past 64 GiB.  The table is one 100 k-query block (Zipf 1..3000 hits per query, deep lineages:
    streamed, long-pass and worklist segments) repeated back to back, so the records must repeat with it —
    reference rows shifted by the block's row count — and the first block is checked against the oracle."""
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    per_row = 20 if layout == "packed" else 24
    if free_b < (1 << 32) * (per_row + 4) + (32 << 30):
        pytest.skip("needs a device with > 130 GB free")
    tax = synth.make_taxonomy(30000, synth.SEEDS["C5"], deep=True)
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=0)
    dh = synth.make_hits(tax, 100_000, synth.SEEDS["C5"], None, zipf=(1.1, 1, 3000), device="cuda")
    Q0, H0 = dh.n_queries, dh.n_hits
    desc_rows = dh.tax_row.clone()
    dh.tax_row = t.engine_rows(desc_rows).contiguous()
    base = dh.as_dict(layout)
    k = ((1 << 32) - 2) // H0
    Hn, Q = k * H0, k * Q0
    assert (1 << 32) - 2 - H0 < Hn < (1 << 32) - 1
    big = {}
    for name, col in base.items():
        if name == "seg_off":
            seg = (col[:-1].to(torch.int64)[None, :] + (torch.arange(k, device="cuda", dtype=torch.int64) * H0)[:, None]).reshape(-1)
            big[name] = torch.cat([seg, torch.tensor([Hn], device="cuda", dtype=torch.int64)])
        else:
            big[name] = col.repeat(k)
    torch.cuda.synchronize()
    print(f"[near-limit] {layout}: {k} blocks of {Q0} queries / {H0} rows = {Q} queries / {Hn} rows", flush=True)
    assert int(big["seg_off"][-1]) == Hn and big["bitscore"].numel() == Hn
    out = torch.zeros(32 * Q, dtype=torch.uint8, device="cuda")
    engine.run_consensus_device(t, big, out, strategy="relaxed")
    torch.cuda.synchronize()
    print("[near-limit] kernels done", flush=True)
    w = out.view(torch.int32).view(k, Q0, 8)                       # word 3 of a record = ref_row
    first = w[0]
    other = torch.ones(8, dtype=torch.bool, device="cuda")
    other[3] = False
    assert bool((w[:, :, other] == first[None, :, other]).all())
    has = first[:, 3] != -1
    ref = w[:, :, 3].to(torch.int64) & 0xFFFFFFFF
    shift = (torch.arange(k, device="cuda", dtype=torch.int64) * H0)[:, None]
    assert bool(((ref - shift)[:, has] == (first[:, 3].to(torch.int64) & 0xFFFFFFFF)[None, has]).all())
    assert bool((w[:, ~has, 3] == -1).all())
    print("[near-limit] periodic", flush=True)
    # the block itself against the oracle
    got = engine.records_from_tensor(out[:32 * Q0])
    sub = {"seg_off": base["seg_off"].cpu().numpy(), "bitscore": dh.bitscore.cpu().numpy(), "tax_row": desc_rows.cpu().numpy(),
           "pident": dh.pident.cpu().numpy(), "align_len": dh.align_len.cpu().numpy(), "acc_rank": dh.acc_rank.cpu().numpy()}
    exp = H.columnar(tax, sub, "bacteria", "relaxed", None, threads=8)
    assert got.tobytes() == exp.tobytes()
    st = got["status"]
    assert (st <= 1).sum() > 0.5 * Q0


def test_one_segment_longer_than_2_31_rows():
    """A single query of 2^31 + 4096 rows between two ordinary ones (the worklist kernel walks it in 2^28-row spans).
    Only the top bit-score group matters, so the record must equal the oracle's for the nine top rows alone, with the
    reference row mapped back to where that row sits in the giant segment — first row, both sides of a span boundary,
    both sides of 2^31, last row.  Second run: the top row past 2^31 has an unmatched taxid (status 16 at that row)."""
    import torch
    free_b, _ = torch.cuda.mem_get_info()
    if free_b < (80 << 30):
        pytest.skip("needs a device with > 80 GB free")
    tax = synth.make_taxonomy(30000, synth.SEEDS["C5"], deep=True)
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=0)
    dh = synth.make_hits(tax, 2000, synth.SEEDS["C5"], 10, device="cuda", p_unmatched=0.0)
    desc = dh.tax_row.clone()
    dh.tax_row = t.engine_rows(desc).contiguous()
    base = dh.as_dict("packed")
    rec0 = base["packed"].view(-1, 4)                               # 20 000 side records
    n_big = (1 << 31) + 4096
    P = torch.tensor([0, 5, (1 << 28) - 1, 1 << 28, (1 << 28) + 7, (1 << 31) - 1, 1 << 31, (1 << 31) + 3, n_big - 1], device="cuda")
    pre, post = 10, 10                                              # query 0 = rows 0..9 of the base, query 2 = rows 10..19
    Hn = pre + n_big + post
    bs = torch.full((Hn,), 100, dtype=torch.int32, device="cuda")
    bs[:pre] = dh.bitscore[:pre]
    bs[pre + n_big:] = dh.bitscore[10:20]
    bs[pre + P] = 200
    reps = (Hn + rec0.shape[0] - 1) // rec0.shape[0]
    rec = rec0.repeat(reps, 1)[:Hn].contiguous()
    rec[:pre] = rec0[:pre]
    rec[pre + n_big:] = rec0[10:20]
    top_src = torch.arange(20, 29, device="cuda")                   # the nine top rows: rows of base query 2 (one neighbourhood)
    rec[pre + P] = rec0[top_src]
    seg = torch.tensor([0, pre, pre + n_big, Hn], dtype=torch.int64, device="cuda")

    def tiny(unmatched_at=None):
        src = top_src.cpu().numpy()
        rows = desc.cpu().numpy()[src].copy()
        if unmatched_at is not None:
            rows[unmatched_at] = -1
        sub = {"seg_off": np.array([0, 9], dtype=np.int64), "bitscore": np.full(9, 200, dtype=np.int32), "tax_row": rows,
               "pident": dh.pident.cpu().numpy()[src], "align_len": dh.align_len.cpu().numpy()[src],
               "acc_rank": dh.acc_rank.cpu().numpy()[src]}
        return H.columnar(tax, sub, "bacteria", "relaxed", None, threads=1)[0]

    def run():
        out = torch.zeros(32 * 3, dtype=torch.uint8, device="cuda")
        engine.run_consensus_device(t, {"seg_off": seg, "bitscore": bs, "packed": rec.view(-1)}, out, strategy="relaxed")
        torch.cuda.synchronize()
        return engine.records_from_tensor(out)

    Pn = P.cpu().numpy()
    for unmatched_at in (None, 7):
        if unmatched_at is not None:
            rec[pre + P[unmatched_at], 0] = -1                       # BLU_UNMATCHED_TAXID
        got = run()
        exp = tiny(unmatched_at).copy()
        assert exp["status"] == (0 if unmatched_at is None else 16)
        exp["ref_row"] = np.uint32(pre + Pn[int(exp["ref_row"])])
        assert got[1].tobytes() == exp.tobytes()
        # the neighbours are ordinary queries
        for qi, (a, b) in ((0, (0, 10)), (2, (10, 20))):
            sub = {"seg_off": np.array([0, 10], dtype=np.int64), "bitscore": dh.bitscore[a:b].cpu().numpy(), "tax_row": desc[a:b].cpu().numpy(),
                   "pident": dh.pident[a:b].cpu().numpy(), "align_len": dh.align_len[a:b].cpu().numpy(), "acc_rank": dh.acc_rank[a:b].cpu().numpy()}
            e = H.columnar(tax, sub, "bacteria", "relaxed", None, threads=1)[0].copy()
            if e["ref_row"] != 0xFFFFFFFF:
                e["ref_row"] = np.uint32(int(e["ref_row"]) + (0 if qi == 0 else pre + n_big))
            assert got[qi].tobytes() == e.tobytes()
