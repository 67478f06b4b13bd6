"""The milli-percent encoding's losslessness claim, exhaustively: for EVERY k the 16-byte side records hold — [0, 131 070] —
the f64 the device rebuilds from k (`milli_to_f64` in consensus_kernel.hip: an IEEE f64 division; `milli17_to_f64` in the
packed and keyed f64 paths: two FMAs around a multiplication by fl(1/1000)) is bit-identical to the host's correctly
rounded k / 1000.0 — the double Rust's `str::parse::<f64>` yields for the 3-decimal text BLAST prints (the reference
parses perc_identity through polars' CSV reader into f64, mod.rs:226-244).  Checked through the C ABI in both the
stream kernel (one-hit queries) and the worklist kernel (segments over 512 rows)."""
import numpy as np
import pytest

from blutils_amd import engine, synth

pytestmark = pytest.mark.gpu

ZERO_CUTS = {"domain": 0, "kingdom": 0, "phylum": 0, "class": 0, "order": 0, "family": 0, "genus": 0, "species": 0}


def test_every_milli_percent_value_converts_exactly():
    tax = synth.make_taxonomy(500, 77)
    # every cutoff 0: any identity passes, so the single-hit record always carries the converted identity
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="custom", custom=ZERO_CUTS, device=0)
    K = np.arange(0, 131071, dtype=np.uint32)
    exp = K.astype(np.float64) / 1000.0                          # numpy's f64 division is IEEE, correctly rounded
    rows = t.engine_rows(np.full(len(K), 3, dtype=np.int32))
    # --- stream kernel: one query per value, one hit each
    seg = np.arange(len(K) + 1, dtype=np.uint64)
    ones = np.ones(len(K), dtype=np.int32)
    for packed in (False, True, "wide"):
        got = engine.run_consensus_host(t, seg, ones * 500, rows, exp if packed == "wide" else None, ones * 400, ones.astype(np.uint32),
                                        strategy="relaxed", pident_milli=None if packed == "wide" else K, packed=packed)
        assert (got["status"] == 1).all()
        assert got["ident_used"].view(np.uint64).tobytes() == exp.view(np.uint64).tobytes()
    # --- worklist kernel: 600-row segments with ONE top row carrying the value (a sample of the range incl. both ends)
    ks = np.unique(np.concatenate([K[::97], K[-3:], K[:3]]))
    n = 600
    seg = (np.arange(len(ks) + 1, dtype=np.uint64) * n)
    bs = np.full(len(ks) * n, 100, dtype=np.int32)
    pm = np.full(len(ks) * n, 12345, dtype=np.uint32)
    top = np.arange(len(ks)) * n + (np.arange(len(ks)) * 7) % n
    bs[top] = 900
    pm[top] = ks
    rows = t.engine_rows(np.full(len(bs), 3, dtype=np.int32))
    ones = np.ones(len(bs), dtype=np.int32)
    got = engine.run_consensus_host(t, seg, bs, rows, None, ones * 400, ones.astype(np.uint32), strategy="cautious",
                                    pident_milli=pm, packed=True)
    assert (got["status"] == 1).all() and (got["ref_row"] == top).all()
    assert got["ident_used"].view(np.uint64).tobytes() == (ks.astype(np.float64) / 1000.0).view(np.uint64).tobytes()
