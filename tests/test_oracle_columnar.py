"""The columnar (integer) oracle must agree with the string-faithful oracle, which is the one pinned on
the reference's golden vectors.  Random synthetic tables, both strategies, built-in and custom backbones."""
import numpy as np
import pytest

from blutils_amd import synth
from oracle import oracle as orc
from tests import helpers as H


def _check(tax, hits, taxon, strategy, custom=None, bad=None):
    recs = H.columnar(tax, hits, taxon, strategy, custom, bad)
    faithful = orc.run(H.oracle_table(tax, hits, bad), taxon=taxon, strategy=strategy, custom=custom, threads=4).results()
    disp, serde, isdef = H.oracle_rank_tables(tax, taxon, custom)
    r = H.Renderer(tax, hits, disp, serde, isdef)
    for q in range(len(recs)):
        H.assert_matches_faithful(r.render(recs[q]), faithful[q], q)
    return recs


@pytest.mark.parametrize("strategy", ["relaxed", "cautious"])
@pytest.mark.parametrize("taxon,custom", [("bacteria", None), ("custom", H.CUSTOM_16S), ("fungi", None)])
def test_columnar_vs_faithful_c1(strategy, taxon, custom):
    tax = synth.make_taxonomy(2000, synth.SEEDS["C1"])
    hits = synth.make_hits(tax, 1000, synth.SEEDS["C1"], 10, p_unmatched=0.002).numpy()
    recs = _check(tax, hits, taxon, strategy, custom)
    st = recs["status"]
    assert (st == 0).sum() > 200 and (st == 1).sum() > 100      # both outcomes exercised


@pytest.mark.parametrize("strategy", ["relaxed", "cautious"])
def test_columnar_vs_faithful_deep_zipf(strategy):
    tax = synth.make_taxonomy(3000, 77, deep=True)
    hits = synth.make_hits(tax, 400, 78, None, zipf=(1.1, 1, 300), p_unmatched=0.001).numpy()
    _check(tax, hits, "eukaryotes", strategy)


def test_columnar_vs_faithful_bad_lineages_and_errors():
    tax = synth.make_taxonomy(500, 5)
    bad = (np.arange(tax.n) % 37 == 0).astype(np.uint8)
    hits = synth.make_hits(tax, 600, 6, 12, p_unmatched=0.01).numpy()
    recs = _check(tax, hits, "bacteria", "relaxed", bad=bad)
    st = set(recs["status"].tolist())
    assert {16, 17, 18}.issubset(st), st                          # unmatched, bad lineage, root disagreement


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("strategy", ["relaxed", "cautious"])
def test_columnar_vs_faithful_adversarial_ties(seed, strategy):
    """Collisions on every sort key, prefix lineages, duplicate lineages, odd rank sequences."""
    tax, hits = H.adversarial_case(seed)
    bad = (np.arange(tax.n) % 17 == 3).astype(np.uint8)
    for taxon, custom in (("bacteria", None), ("custom", H.CUSTOM_16S)):
        recs = _check(tax, hits, taxon, strategy, custom, bad)
    assert len(set(recs["status"].tolist())) >= 5
