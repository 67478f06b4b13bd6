"""Randomised sweep: taxonomies of different depth, segment-length mixes that hit every path of the stream kernel (dense
and two-stage steps of every width, the long pass, the worklist kernel), both strategies, the three hit-table layouts,
built-in and custom backbones — every record against the columnar oracle."""
import numpy as np
import pytest

from blutils_amd import engine, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu

MIXES = {
    "tiny": lambda rng, n: rng.integers(1, 9, n),
    "short": lambda rng, n: rng.integers(1, 25, n),
    "c3like": lambda rng, n: rng.integers(40, 61, n),
    "mid": lambda rng, n: rng.integers(60, 140, n),
    "long": lambda rng, n: rng.integers(120, 520, n),
    "mixed": lambda rng, n: rng.choice([0, 1, 3, 10, 17, 33, 64, 65, 100, 129, 257, 400, 513, 700], n),
    "heavy_tail": lambda rng, n: np.minimum(rng.zipf(1.3, n), 1500),
}


@pytest.mark.parametrize("seed", range(21))
def test_random_tables_against_the_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    mix = list(MIXES)[seed % len(MIXES)]
    deep = bool(seed % 2)
    tax = synth.make_taxonomy(int(rng.integers(500, 20000)), 50 + seed, deep=deep)
    n_q = 1500 if mix in ("long", "mixed", "heavy_tail", "mid") else 4000
    lens = np.asarray(MIXES[mix](rng, n_q), dtype=np.int64)
    cap = int(max(lens.max(), 1))
    base = synth.make_hits(tax, n_q, 70 + seed, cap, p_unmatched=0.002 if seed % 3 == 0 else 0.0)
    hb = base.numpy()
    milli_full = base.pident_milli.numpy()
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    take = np.concatenate([np.arange(l) + cap * i for i, l in enumerate(lens)]) if lens.sum() else np.zeros(0, np.int64)
    h = {k: (hb[k][take] if k != "seg_off" else seg) for k in hb}
    milli = milli_full[take]
    taxon, custom = (("custom", H.CUSTOM_16S) if seed % 2 else (["bacteria", "fungi", "eukaryotes"][seed % 3], None))
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon=taxon, custom=custom, device=0, taxid=tax.taxid)
    rows = t.engine_rows(h["tax_row"])
    for strategy in ("relaxed", "cautious"):
        exp = H.columnar(tax, h, taxon, strategy, custom)
        f64 = engine.run_consensus_host(t, h["seg_off"], h["bitscore"], rows, h["pident"], h["align_len"], h["acc_rank"], strategy)
        assert f64.tobytes() == exp.tobytes(), (mix, strategy, "f64")
        col = engine.run_consensus_host(t, h["seg_off"], h["bitscore"], rows, None, h["align_len"], h["acc_rank"], strategy, pident_milli=milli)
        assert col.tobytes() == exp.tobytes(), (mix, strategy, "milli")
        pk = engine.run_consensus_host(t, h["seg_off"], h["bitscore"], rows, None, h["align_len"], h["acc_rank"], strategy, pident_milli=milli,
                                       packed=True)
        assert pk.tobytes() == exp.tobytes(), (mix, strategy, "packed")
