"""Pin the CPU oracle on every golden vector the reference tree holds."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_recipe import table_from_taxa


@pytest.fixture(scope="module")
def zymo(golden_dir):
    with gzip.open(os.path.join(golden_dir, "zymo_mock_distilled.json.gz"), "rt") as f:
        return json.load(f)


CORE = ("singleMatch", "reachedRank", "identifier", "taxonomy", "percIdentity", "bitScore")


@pytest.mark.parametrize("strategy", ["relaxed", "cautious"])
def test_zymo_golden_core_fields(zymo, strategy):
    """(singleMatch, reachedRank, identifier, taxonomy) reproduce for 2283/2283 results."""
    taxa = [c["taxon"] for c in zymo["cases"]]
    weights = [c["n_queries"] for c in zymo["cases"]]
    assert sum(weights) == 2283 and len(taxa) == 253
    run = orc.run(table_from_taxa(taxa), taxon=zymo["config"]["taxon"], strategy=strategy)
    got = run.results()
    n_full = 0
    for exp, g, w in zip(taxa, got, weights):
        assert g["status"] == orc.ST_CONSENSUS
        for k in CORE:
            assert g["taxon"][k] == exp[k], (k, exp["identifier"])
        if g["taxon"]["maxAllowedRank"] == exp["maxAllowedRank"] and g["taxon"]["mutated"] == exp["mutated"]:
            n_full += w
    # 1856/2283 under relaxed, 1626 under cautious also give back
    # maxAllowedRank+mutated (the rest lost the reference row's deeper ranks in the bean fold).
    assert n_full == {"relaxed": 1856, "cautious": 1626}[strategy]


@pytest.mark.parametrize("reverse", [False, True])
def test_zymo_golden_beans(zymo, reverse):
    """Folded beans (rank, identifier, occurrences, taxonomy, accessions) come back in the golden order — with the hit rows of
    every query in the recipe's file order and in the opposite one (the sort, not the file, decides)."""
    taxa = [c["taxon"] for c in zymo["cases"]]
    n_multi = sum(1 for t in taxa for b in t["consensusBeans"] if len(b["accessions"]) > 1)
    n_unsorted = sum(1 for t in taxa for b in t["consensusBeans"] if b["accessions"] != sorted(b["accessions"]))
    assert n_multi > 300 and n_unsorted > 50      # (distinct taxon objects; over the 2283 results: 3586 and 585)
    got = orc.run(table_from_taxa(taxa, reverse_file_order=reverse), taxon="bacteria", strategy="relaxed").results()
    for exp, g in zip(taxa, got):
        eb, gb = exp["consensusBeans"], g["taxon"]["consensusBeans"]
        assert len(eb) == len(gb)
        for a, b in zip(eb, gb):
            assert a["rank"] == b["rank"] and a["identifier"] == b["identifier"]
            assert a["occurrences"] == b["occurrences"]
            assert a["taxonomy"] == b["taxonomy"]
            # the accessions of a bean in the golden's own order: the recipe's ascending align_length makes that order what
            # the 4-key stable sort has to produce (585 of the 3586 multi-accession beans are not in ascending accession order)
            assert a["accessions"] == b["accessions"]


def test_zymo_single_match_kats(zymo):
    """The singleMatch:true results are exact single-hit known-answer tests (all nine fields)."""
    taxa = [c["taxon"] for c in zymo["cases"] if c["taxon"]["singleMatch"]]
    assert sum(c["n_queries"] for c in zymo["cases"] if c["taxon"]["singleMatch"]) == 30
    got = orc.run(table_from_taxa(taxa), taxon="bacteria", strategy="relaxed").results()
    for exp, g in zip(taxa, got):
        assert g["taxon"] == exp


def test_docs_worked_example(golden_dir):
    """docs/book/02_*.md:192-249 (8.3.1, bacteria, relaxed) — every field of the taxon object."""
    doc = json.load(open(os.path.join(golden_dir, "docs_worked_example.json")))
    taxa = [r["taxon"] for r in doc["results"]]
    got = orc.run(table_from_taxa(taxa), taxon=doc["taxon"], strategy=doc["strategy"]).results()
    for exp, g in zip(taxa, got):
        assert g["status"] == orc.ST_CONSENSUS
        assert g["taxon"] == exp


def test_cutoff_tables_and_interpolation_examples(golden_dir):
    """taxon.rs:144-184 tables + the interpolation cases worked out in SURVEY §3.3."""
    full = ["d", "k", "p", "c", "o", "f", "g", "s"]
    c, d = orc.interpolate(full, "bacteria")
    # bacteria backbone has no Kingdom (taxon.rs:159-169): k is interpolated over the window [d,k,p]
    np.testing.assert_array_equal(c, [60.0, 67.5, 75.0, 80.0, 85.0, 92.0, 97.0, 99.0])
    assert list(d) == [True, False, True, True, True, True, True, True]
    c, _ = orc.interpolate(["d", "p", "c", "o", "f", "g", "s"], "fungi")
    np.testing.assert_array_equal(c, [60.0, 75.0, 80.0, 85.0, 90.0, 95.0, 97.0])
    c, _ = orc.interpolate(["d", "p", "c", "o", "f", "g", "s"], "eukaryotes")
    np.testing.assert_array_equal(c, [60.0, 75.0, 80.0, 85.0, 90.0, 95.0, 97.0])
    c, _ = orc.interpolate(["d", "clade", "p", "c", "o", "f", "g", "s"], "bacteria")
    assert c[1] == 67.5
    c, _ = orc.interpolate(["d", "p", "c", "o", "f", "g", "s", "strain"], "bacteria")
    assert c[-1] == 100.0
    c, _ = orc.interpolate(
        ["cellular-root", "d", "k", "p", "c", "o", "f", "g", "species-group", "species-subgroup", "s"], "bacteria")
    np.testing.assert_array_equal(c, [99.0, 60.0, 66.667, 75.0, 80.0, 85.0, 92.0, 97.0, 97.667, 98.333, 99.0])
    c, _ = orc.interpolate(["no-rank", "superkingdom", "p", "c", "o", "f", "g", "s"], "bacteria")
    np.testing.assert_array_equal(c[:3], [99.0, 87.0, 75.0])
    custom = json.load(open(os.path.join(golden_dir, "custom_taxon_cutoffs_bacteria_16S.json")))["values"]
    c, d = orc.interpolate(["cellular-root"] + full, "custom", custom)
    np.testing.assert_array_equal(c, [50.0, 50.0, 60.0, 75.0, 80.0, 85.0, 92.0, 97.0, 99.0])
    assert list(d) == [False] + [True] * 8
    # one-element window: 0 * (w / 0) = NaN by IEEE rules, not a panic
    c, _ = orc.interpolate(["clade"], "bacteria")
    assert np.isnan(c[0])


def test_rank_parsing():
    """linnaean_ranks.rs:52-89: letters and full names map to the enum, the rest to Other(slug)."""
    assert orc.rank_display("Domain") == "d" and orc.rank_serde("d") == "domain"
    assert orc.rank_display(" S ") == "s" and orc.rank_serde("species") == "species"
    assert orc.rank_display("u") == "u" and orc.rank_serde("u") == "undefined"
    assert orc.rank_display("species-group") == "species-group"
    assert orc.rank_serde("Cellular Root") == "cellular-root"
    assert orc.rank_serde("no rank") == "no-rank"
