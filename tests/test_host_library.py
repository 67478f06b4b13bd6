"""CPU-side checks of the product library: it loads, exports every symbol of include/blu_consensus.h,
refuses to compute without a GPU, and its per-shape cutoff tables (host C++ in blutils_amd/csrc/taxonomy.cpp)
equal the oracle's restatement of InterpolatedIdentity::interpolate_identities."""
import os
import re

import numpy as np
import pytest

from blutils_amd import _native as N
from blutils_amd import engine, synth
from oracle import oracle as orc
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_driver_build_hook_runs():
    """__graft_entry__.build() is what the driver calls to check that everything compiles (here, without a GPU): an
    incremental `make` of the library and of the oracle, then the ABI version of the header against the library's."""
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "blu_consensus.h")).read()
    declared = set(re.findall(r"\b(blu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(N.EXPORTS)
    L = N.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.blu_abi_version() == 5
    hdr2 = open(os.path.join(ROOT, "include", "blu_pipeline.h")).read()
    declared2 = set(re.findall(r"\b(blu_[a-z0-9_]+)\s*\(", hdr2))
    assert declared2 == set(N.PIPELINE_EXPORTS)
    for name in declared2:
        assert hasattr(L, name), name


def test_custom_taxon_from_file(tmp_path, golden_dir):
    """CustomTaxon::from_file (taxon.rs:28-66): the reference's assets YAML (as JSON fixture + re-written YAML)."""
    import json
    from blutils_amd import pipeline
    vals = json.load(open(os.path.join(golden_dir, "custom_taxon_cutoffs_bacteria_16S.json")))["values"]
    y = tmp_path / "cutoffs.yaml"
    y.write_text("".join(f"{k}: {v}\n" for k, v in vals.items()))
    assert pipeline.custom_taxon_from_file(str(y)) == vals
    j = tmp_path / "cutoffs.json"
    j.write_text(json.dumps({"domain": 55, "kingdom": None, "species": 98}))
    assert pipeline.custom_taxon_from_file(str(j)) == {"domain": 55, "species": 98}
    with pytest.raises(N.BluError):
        pipeline.custom_taxon_from_file(str(tmp_path / "cutoffs.txt"))


def test_pipeline_needs_a_device(tmp_path):
    """The drop-in use-case has no CPU path either."""
    import json
    import torch
    from blutils_amd import pipeline
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    (tmp_path / "t.json").write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": 1, "rank": "species", "numericLineage": "d__2;s__1", "textLineage": "d__b;s__x", "accessions": []}]}))
    (tmp_path / "b.tsv").write_text("q1\tA.1\t1\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n")
    with pytest.raises(N.BluError) as e:
        pipeline.build_consensus_identities(str(tmp_path / "b.tsv"), str(tmp_path / "t.json"))
    assert e.value.code == N.BLU_ERR_NO_DEVICE


def test_result_record_layout():
    assert engine.RESULT_DTYPE.itemsize == 32
    assert [engine.RESULT_DTYPE.fields[k][1] for k in engine.RESULT_DTYPE.names] == [0, 1, 2, 3, 4, 6, 8, 12, 16, 24]


def _host_tax(tax, taxon, custom=None):
    return engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon=taxon, custom=custom,
                           device=-1, taxid=tax.taxid)


@pytest.mark.parametrize("taxon,custom", [("bacteria", None), ("fungi", None), ("eukaryotes", None),
                                          ("custom", H.CUSTOM_16S), ("custom", {"domain": 55, "species": 98})])
@pytest.mark.parametrize("deep", [False, True])
def test_shape_cutoffs_match_oracle(taxon, custom, deep):
    tax = synth.make_taxonomy(4000, 11, deep=deep)
    t = _host_tax(tax, taxon, custom)
    seen = {}
    for row in range(tax.n):
        a, b = int(tax.lin_off[row]), int(tax.lin_off[row + 1])
        key = tuple(tax.lin_rank[a:b])
        if key in seen:
            continue
        seen[key] = row
        cut, isdef, codes = t.row_cutoffs(row)
        ocut, oisdef = orc.interpolate([tax.rank_names[i] for i in key], taxon, custom)
        np.testing.assert_array_equal(cut.view(np.uint64), ocut.view(np.uint64))   # bit-exact, NaN included
        assert list(isdef) == list(oisdef)
        for j, i in enumerate(key):
            assert t.rank_name(codes[j]) == orc.rank_display(tax.rank_names[i])
            assert t.rank_name(codes[j], serde=True) == orc.rank_serde(tax.rank_names[i])
    assert t.n_shapes == len(seen) and len(seen) > (50 if deep else 10)


def test_pathological_rank_sequences_match_oracle():
    """duplicate ranks, leading/trailing non-default runs, one-element windows (NaN), mixed-case names."""
    rng = np.random.default_rng(3)
    names = ["d", "Kingdom", "p", "c", "o", "f", "g", "s", "clade", "no rank", "strain", "u", "Domain", "clade "]
    seqs = [["clade"], ["clade", "d", "clade"], ["strain", "strain"], ["d", "d", "clade", "d"], ["u"], ["s", "clade", "d"]]
    for _ in range(300):
        n = int(rng.integers(1, 14))
        seqs.append([names[int(i)] for i in rng.integers(0, len(names), n)])
    off = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    rank = np.array([names.index(r) for s in seqs for r in s], dtype=np.uint16)
    node = np.arange(len(rank), dtype=np.uint32)
    for taxon, custom in (("bacteria", None), ("custom", H.CUSTOM_16S)):
        t = engine.Taxonomy(off, node, rank, names, taxon=taxon, custom=custom, device=-1)
        for i, s in enumerate(seqs):
            cut, isdef, _ = t.row_cutoffs(i)
            ocut, oisdef = orc.interpolate(s, taxon, custom)
            np.testing.assert_array_equal(cut.view(np.uint64), ocut.view(np.uint64), err_msg=str(s))
            assert list(isdef) == list(oisdef)


def test_custom_taxon_without_values_is_an_error():
    tax = synth.make_taxonomy(50, 1)
    with pytest.raises(N.BluError) as e:
        _host_tax(tax, "custom", None)
    assert e.value.code == N.BLU_ERR_CUSTOM_MISSING      # taxon.rs:117 panic -> call-level error


def test_too_deep_lineage_is_refused():
    off = np.array([0, 65], dtype=np.uint64)
    with pytest.raises(N.BluError) as e:
        engine.Taxonomy(off, np.arange(65, dtype=np.uint32), np.zeros(65, dtype=np.uint16), ["clade"], device=-1)
    assert e.value.code == N.BLU_ERR_DEPTH


def test_taxid_lookup():
    tax = synth.make_taxonomy(300, 2)
    t = _host_tax(tax, "bacteria")
    rows = t.lookup(np.array([tax.taxid[5], 999999999, tax.taxid[299]], dtype=np.int64))
    fwd, inv = t.row_map()
    assert rows.tolist() == [int(fwd[5]), N.BLU_UNMATCHED_TAXID, int(fwd[299])]     # engine row ids
    pos, length = fwd & ((1 << 25) - 1), fwd >> 25
    assert sorted(pos.tolist()) == list(range(tax.n)) and (inv[pos] == np.arange(tax.n)).all()
    assert (length == np.diff(tax.lin_off.astype(np.int64))).all()                  # the id carries the lineage length
    # engine row ids follow the lexicographic order of the lineages
    lin = [tuple(tax.lin_node[int(tax.lin_off[i]):int(tax.lin_off[i + 1])]) for i in inv]
    assert lin == sorted(lin)
    assert t.engine_rows(np.array([5, -1, 299], dtype=np.int32)).tolist() == rows.tolist()


def test_no_cpu_fallback():
    """Without a device the hot path must fail loudly, never compute on the host."""
    tax = synth.make_taxonomy(100, 1)
    hits = synth.make_hits(tax, 10, 2, 5).numpy()
    t = _host_tax(tax, "bacteria")
    with pytest.raises(N.BluError) as e:
        engine.run_consensus_host(t, hits["seg_off"], hits["bitscore"], hits["tax_row"], hits["pident"],
                                  hits["align_len"], hits["acc_rank"])
    assert e.value.code == N.BLU_ERR_NO_DEVICE
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(N.BluError) as e2:
            engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, device=0)
        assert e2.value.code == N.BLU_ERR_NO_DEVICE


def test_shard_ranges_match_the_python_sharding():
    """blu_shard_ranges (C ABI, used by blu_consensus_run_multi) cuts like blutils_amd.shard.balanced_query_ranges."""
    from blutils_amd import engine, shard
    rng = np.random.default_rng(4)
    for trial in range(40):
        nq = int(rng.integers(1, 400))
        counts = rng.integers(0, 60, nq) if trial % 3 else (rng.zipf(1.3, nq) % 3000)
        seg = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
        for parts in (1, 2, 3, 8, 13):
            got = engine.shard_ranges(seg, parts)
            exp = shard.balanced_query_ranges(seg, parts)
            assert [(int(got[i]), int(got[i + 1])) for i in range(parts)] == exp
            assert got[0] == 0 and got[-1] == nq and np.all(np.diff(got.astype(np.int64)) >= 0)


def test_sharded_run_validates_the_offset_table_before_touching_a_device():
    """blu_consensus_run_multi cuts the host table by its offsets: a non-ascending table, or one that runs past n_hits,
    is refused as data (BLU_ERR_INVALID_ARG) — with host-only handles, i.e. before any device is asked for."""
    import ctypes as C
    tax = synth.make_taxonomy(100, 1)
    a, b = _host_tax(tax, "bacteria"), _host_tax(tax, "bacteria")
    h = synth.make_hits(tax, 10, 2, 5).numpy()
    L = N.lib()
    L.blu_consensus_run_multi.restype = C.c_int
    L.blu_consensus_run_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    out = np.zeros(10, dtype=engine.RESULT_DTYPE)
    params = N.RunParams(N.STRATEGY["relaxed"], 0, None)
    handles = (C.c_void_p * 2)(a.handle, b.handle)
    cols = [np.ascontiguousarray(h[k]) for k in ("bitscore", "tax_row", "pident", "align_len", "acc_rank")]

    def run(seg, nh, hs=handles, n=2):
        seg = np.ascontiguousarray(seg, dtype=np.uint64)
        hits = N.Hits(cols[0].ctypes.data, cols[1].ctypes.data, cols[2].ctypes.data, cols[3].ctypes.data, cols[4].ctypes.data,
                      seg.ctypes.data, nh, len(seg) - 1, 0, 0, None, None)
        return L.blu_consensus_run_multi(hs, n, C.byref(hits), C.byref(params), out.ctypes.data)

    seg = h["seg_off"].astype(np.uint64)
    assert run(seg, 40) == N.BLU_ERR_INVALID_ARG                      # offsets run to 50, the table holds 40 rows
    bad = seg.copy()
    bad[3], bad[4] = bad[4], bad[3]
    assert run(bad, 50) == N.BLU_ERR_INVALID_ARG                      # not ascending
    assert run(seg, 50, (C.c_void_p * 2)(a.handle, a.handle)) == N.BLU_ERR_INVALID_ARG   # one handle twice
    assert run(seg, 50) == N.BLU_ERR_NO_DEVICE                        # well-formed: only now a device is needed


def test_hits_pack_words_and_shape_hints():
    """blu_hits_pack / blu_hits_pack64 on host arrays (no GPU): the four values of a hit side by side, and in the bits above
    the 17-bit identity a hint that is a function of the row's rank sequence — the same for rows of one shape, different
    for rows of different shapes (it is `shape id + 1`), 0 for an unmatched row; identities the 16-byte record cannot hold
    are refused."""
    tax = synth.make_taxonomy(3000, 23)
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=-1)
    rng = np.random.default_rng(1)
    n = 5000
    desc = rng.integers(0, tax.n, n).astype(np.int32)
    desc[::97] = -1                                                    # unmatched
    rows = t.engine_rows(desc)
    pm = rng.integers(0, 131071, n).astype(np.uint32)
    aln = rng.integers(-5, 3000, n).astype(np.int32)
    acc = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    rec = engine.pack_records(t, rows, pm, aln, acc)
    assert rec.shape == (n, 4)
    assert np.array_equal(rec[:, 0], rows.view(np.uint32)) and np.array_equal(rec[:, 1] & 0x1FFFF, pm)
    assert np.array_equal(rec[:, 2], aln.view(np.uint32)) and np.array_equal(rec[:, 3], acc)
    hint = rec[:, 1] >> 17
    shape_of = {}
    for i in range(n):
        if desc[i] < 0:
            assert hint[i] == 0
            continue
        a, b = int(tax.lin_off[desc[i]]), int(tax.lin_off[desc[i] + 1])
        key = tuple(tax.rank_names[r] for r in tax.lin_rank[a:b])
        assert hint[i] != 0
        assert shape_of.setdefault(key, int(hint[i])) == int(hint[i])
    assert len(set(shape_of.values())) == len(shape_of) > 20           # one hint per shape
    assert t.n_shapes >= len(shape_of) and max(shape_of.values()) <= t.n_shapes
    # the wide records carry the f64 as it is
    pid = pm.astype(np.float64) / 1000.0
    pid[5] = np.nan; pid[6] = 1e308
    wide = engine.pack_records(t, rows, None, aln, acc, pident=pid, wide=True)
    assert wide.shape == (n, 6) and np.array_equal(wide[:, 1] >> 17, hint) and np.array_equal(wide[:, 1] & 0x1FFFF, np.zeros(n, np.uint32))
    assert np.array_equal(wide[:, 4:6].copy().view(np.uint64).reshape(-1), pid.view(np.uint64))
    # an exact milli-percent f64 column packs into the 16-byte records; anything else, or 131.071 and up, does not
    ok = engine.pack_records(t, rows, None, aln, acc, pident=pm.astype(np.float64) / 1000.0)
    assert np.array_equal(ok, rec)
    for bad in (np.float64(97.0001), np.float64(131.071), np.float64(-1.0), np.nan):
        p2 = pm.astype(np.float64) / 1000.0
        p2[77] = bad
        with pytest.raises(N.BluError, match="milli-percent"):
            engine.pack_records(t, rows, None, aln, acc, pident=p2)
    big = pm.copy(); big[3] = 131071
    with pytest.raises(N.BluError, match="milli-percent"):
        engine.pack_records(t, rows, big, aln, acc)


def test_division_free_milli_percent_conversion_is_exact_for_17_bit_values():
    """consensus_kernel.hip `milli17_to_f64`: with y = fl(1/1000), q = fl(k y), r = fma(-q, 1000, k), the value fma(r, y, q) is
    the correctly rounded k / 1000 for every k below 2^17 — here in exact rational arithmetic (a Fraction converts to the
    nearest double), on the device in tests/test_gpu_milli_exact.py."""
    from fractions import Fraction
    y = Fraction(0.001)
    for k in range(0, 1 << 17):
        q = float(Fraction(k) * y)
        r = float(Fraction(k) - Fraction(q) * 1000)
        assert float(Fraction(q) + Fraction(r) * y) == k / 1000.0, k


def test_worklist_queues_fit_the_buffer_the_library_allocates():
    """consensus_kernel.hip: 64 worklist queues (WL_QUEUES), a task appends to queue task % 64, at most its 64 queries; a queue
    holds wl_capacity(n_queries) = ceil(n_tasks / 64) * 64 entries and the 64 of them lie back to back in the buffer api.cpp
    allocates (n_queries + 8192 words).  The arithmetic, for small, odd and huge query counts."""
    for nq in [0, 1, 63, 64, 65, 4095, 4096, 4097, 10**6, 10**7, 10**7 + 1, 2**32 - 2]:
        n_tasks = (nq + 63) // 64
        cap = (n_tasks + 63) // 64 * 64
        for s_ in (0, 1, 63):
            tasks_of_queue = len(range(s_, n_tasks, 64))
            assert tasks_of_queue * 64 <= cap
        assert 64 * cap <= nq + 8192


def test_wide_node_tables_agree_with_the_adjacent_row_scan():
    """Levels shared by a span of sorted rows: the wide-node chains (spans of 128 rows and more, what phase 2c of the stream
    kernel reads) and the range-minimum tables must both give what a row-by-row scan of the adjacent-row prefix lengths gives
    (find_multi_taxa_consensus.rs:137-180 seen in sorted order)."""
    import ctypes as C
    from blutils_amd import synth
    rng = np.random.default_rng(5)
    for n, deep in ((30000, False), (30000, True), (300, False), (129, True)):
        tx = synth.make_taxonomy(n, 77 + n, deep=deep)
        t = engine.Taxonomy(tx.lin_off, tx.lin_node, tx.lin_rank, tx.rank_names, taxon="bacteria", device=-1)
        L = N.lib()
        scan, tab, via = C.c_uint32(), C.c_uint32(), C.c_int32()
        n_chain = 0
        # every span width class: narrow, just wide, a few blocks, most of the table; spans that start / end on block edges
        los = np.concatenate([rng.integers(0, n, 3000), np.arange(0, min(n, 400)), (np.arange(0, n, 64))[:400], np.maximum(np.arange(0, n, 64) - 1, 0)[:400]])
        for lo in los:
            for w in (0, 1, 5, 127, 128, 129, 200, 1000, int(rng.integers(1, n))):
                hi = min(n - 1, int(lo) + w)
                assert L.blu_taxonomy_shared_levels(t.handle, int(lo), hi, C.byref(scan), C.byref(tab), C.byref(via)) == N.BLU_OK
                assert scan.value == tab.value, (n, deep, int(lo), hi, scan.value, tab.value, via.value)
                n_chain += via.value
        if n >= 30000:
            assert n_chain > 1000          # the chains, not only the fallback, were exercised
        t.close()
