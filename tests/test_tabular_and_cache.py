"""`build-tabular` (parse_consensus_as_tabular/mod.rs:15-173) and the binary taxonomy cache.
Host-only: runs without a GPU."""
import io
import json
import os

import numpy as np
import pytest

from blutils_amd import _native as N
from blutils_amd import cli, pipeline, synth, tabular

RUN = "6f9619ff-8b86-d011-b42d-00c04fc964ff"
DOC = {"results": [
    {"runId": RUN, "query": "q1", "taxon": {
        "reachedRank": "genus", "maxAllowedRank": None, "identifier": "ba", "percIdentity": 99.5, "bitScore": 700.0,
        "taxonomy": "d__bacteria;g__ba", "mutated": False, "singleMatch": False, "consensusBeans": [
            {"rank": "species", "identifier": "ba-x", "occurrences": 2, "taxonomy": "d__bacteria;g__ba;s__ba-x",
             "accessions": ["ACC_A.1", "ACC_C.1"]},
            {"rank": "species-subgroup", "identifier": "ba-y", "occurrences": 1, "taxonomy": None, "accessions": []}]}},
    {"runId": RUN, "query": "q2", "taxon": None},
    {"runId": None, "query": "q3", "taxon": {
        "reachedRank": "clade", "maxAllowedRank": "species", "identifier": "c1", "percIdentity": 100.0, "bitScore": 1e21,
        "taxonomy": None, "mutated": True, "singleMatch": True, "consensusBeans": None}}],
    "config": None}

ROWS = ["\t".join(tabular.HEADER),
        f"{RUN}\tq1\tconsensus\tgenus\tba\t99.5\t700\td__bacteria;g__ba\tfalse\tfalse\tnull\tnull",
        f"{RUN}\tq1\tblast-match\tspecies\tba-x\tnull\t700\td__bacteria;g__ba;s__ba-x\tnull\tnull\t2\tACC_A.1, ACC_C.1",
        f"{RUN}\tq1\tblast-match\tspecies-subgroup\tba-y\tnull\t700\tnull\tnull\tnull\t1\t",
        "q2\tnull\n"]


def test_rust_float_display():
    for v, s in ((99.5, "99.5"), (700.0, "700"), (1e21, "1000000000000000000000"), (1e-7, "0.0000001"), (0.1 + 0.2, "0.30000000000000004"),
                 (-0.0, "-0"), (float("nan"), "NaN"), (float("inf"), "inf"), (66.667, "66.667")):
        assert tabular.rust_f64(v) == s


@pytest.mark.parametrize("fmt", ["json", "jsonl", "yaml"])
def test_build_tabular_stdout_and_file(tmp_path, fmt):
    src = tmp_path / f"res.{fmt}"
    if fmt == "json":
        src.write_text(json.dumps(DOC, indent=2))
    elif fmt == "jsonl":
        # (with the `null` config line of a config-less run in front, the reference's reader fails — as does this one)
        src.write_text("null\n" + "\n".join(json.dumps(r) for r in DOC["results"]) + "\n")
        (tmp_path / "res.json").write_text("")
        with pytest.raises(tabular.TabularError, match="unable to parse line as JSON"):
            tabular.parse_consensus_as_tabular(str(src), None, fmt, stdout=io.StringIO())
        os.remove(tmp_path / "res.json")
        src.write_text("\n".join(json.dumps(r) for r in DOC["results"]) + "\n")
    else:
        import yaml
        src.write_text(yaml.safe_dump(DOC))
    if fmt != "json":
        # the existence check probes the `.json` sibling whatever the format (mod.rs:24-32)
        with pytest.raises(tabular.TabularError, match="does not exist"):
            tabular.parse_consensus_as_tabular(str(src), None, fmt, stdout=io.StringIO())
        (tmp_path / "res.json").write_text("")
    out = io.StringIO()
    tabular.parse_consensus_as_tabular(str(src), None, fmt, stdout=out)
    lines = out.getvalue().split("\n")
    assert lines[:4] == ROWS[:4]
    assert lines[4:6] == ["q2\tnull", ""]                     # the row carries its own newline, println! adds one
    q3 = lines[6].split("\t")
    assert len(q3[0]) == 36 and q3[0] != RUN                   # no runId, no config: a fresh UUID v4 for the call
    assert q3[1:] == ["q3", "consensus", "clade", "c1", "100", "1000000000000000000000", "null", "true", "true", "null", "null"]
    assert lines[7:] == [""]
    # to a file: extension forced to .tsv, an existing file is replaced, rows are appended without terminators
    target = tmp_path / "table.txt"
    (tmp_path / "table.tsv").write_text("stale")
    tabular.parse_consensus_as_tabular(str(src), str(target), fmt)
    text = (tmp_path / "table.tsv").read_text()
    assert text.startswith("".join(ROWS)) and text.endswith("\ttrue\ttrue\tnull\tnull") and text.count("\n") == 1


def test_build_tabular_cli_and_config_run_id(tmp_path, capsys):
    doc = {"results": [dict(DOC["results"][2])], "config": {"runId": RUN, "isConfig": True}}
    (tmp_path / "r.json").write_text(json.dumps(doc))
    assert cli.main(["blastn", "build-tabular", str(tmp_path / "r.json")]) == 0
    assert capsys.readouterr().out.split("\n")[1].startswith(RUN + "\tq3\tconsensus\tclade")
    (tmp_path / "r.jsonl").write_text(json.dumps(doc["config"]) + "\n\n" + json.dumps(doc["results"][0]) + "\n")
    assert cli.main(["blastn", "build-tabular", str(tmp_path / "r.jsonl"), "-i", "jsonl"]) == 0
    assert capsys.readouterr().out.split("\n")[1].startswith(RUN + "\tq3\t")
    with pytest.raises(SystemExit):
        cli.main(["blastn", "build-tabular", str(tmp_path / "absent.json")])
    (tmp_path / "bad.json").write_text("{")
    with pytest.raises(SystemExit, match="unable to parse content as JSON"):
        cli.main(["blastn", "build-tabular", str(tmp_path / "bad.json")])


def _write_db(tmp_path, n=3000, duplicate=False):
    tax = synth.make_taxonomy(n, 11)
    lt, ln = tax.lineage_strings(text=True), tax.lineage_strings(text=False)
    lt[5] = "d__bacteria;broken"                                   # a lineage that fails parse_taxonomy
    db = {"blutilsVersion": "8.3.1", "sourceDatabase": "synthetic", "taxonomies": [
        {"taxid": int(tax.taxid[t]), "rank": "species", "numericLineage": ln[t], "textLineage": lt[t], "accessions": []}
        for t in range(n)]}
    if duplicate:
        db["taxonomies"].append(dict(db["taxonomies"][7], textLineage="d__other"))
    tj = tmp_path / "tax.blutils.json"
    tj.write_text(json.dumps(db))
    rng = np.random.default_rng(3)
    rows = []
    for q in range(400):
        for _ in range(int(rng.integers(1, 12))):
            t = int(rng.integers(0, n))
            taxid = int(tax.taxid[t]) if rng.random() > 0.01 else 987654321
            rows.append(f"q{q:05d}\tNR_{t:06d}.1\t{taxid}\t{rng.integers(80000, 100001) / 1000:.3f}\t{int(rng.integers(380, 480))}"
                        f"\t0\t0\t1\t400\t1\t400\t1e-50\t{int(rng.integers(200, 900))}")
    bt = tmp_path / "blast.tsv"
    bt.write_text("\n".join(rows) + "\n")
    return str(bt), str(tj)


def test_a_taxid_listed_twice_multiplies_its_hit_rows_like_the_left_join(tmp_path):
    """The reference joins hits and taxonomies with polars (mod.rs:72-76, a left join): a taxid the DB lists m times gives m
    joined rows per hit of that subject, in the DB's order.  The ingest does the same — from the JSON and from the binary
    cache — checked column by column against the independent reading (tests/ingest_reference.py); the strict number
    parsing of the Int64 columns ("12.7" as align_length, "12abc" as a score) is checked below."""
    from tests import ingest_reference as ref
    bt, tj = _write_db(tmp_path, duplicate=True)
    exp = ref.read_table(bt, tj)
    (tmp_path / "plain").mkdir()
    plain = ref.read_table(bt, _write_db(tmp_path / "plain", duplicate=False)[1])
    assert len(exp["bitscore"]) > len(plain["bitscore"])                  # the duplicated subject has hits
    got = pipeline.ingest_columns(bt, tj, device=-1)
    for k in ("seg_off", "bitscore", "align_len", "tax_desc_row", "acc_rank"):
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(got["pident"].view(np.uint64), exp["pident"].view(np.uint64))
    cache = str(tmp_path / "dup.blucache")
    cli.main(["cache-db", tj, cache])
    st, ck = pipeline.ingest_only(bt, cache, False)
    assert ck == ref.checksum(exp) and st["n_hits"] == len(exp["bitscore"])
    bt, tj = _write_db(tmp_path)
    good = open(bt).read().splitlines()
    cols = good[3].split("\t")
    for col, value in ((4, cols[4] + ".7"), (2, cols[2] + "e0"), (12, cols[12] + "abc"), (3, " " + cols[3])):
        c = list(cols)
        c[col] = value
        (tmp_path / "bad.tsv").write_text("\n".join(good[:3] + ["\t".join(c)] + good[4:]) + "\n")
        with pytest.raises(N.BluError, match="does not parse"):
            pipeline.ingest_only(str(tmp_path / "bad.tsv"), tj, False)


@pytest.mark.parametrize("use_taxid", [False, True])
def test_db_cache_gives_the_same_ingest(tmp_path, use_taxid):
    bt, tj = _write_db(tmp_path)
    cache = str(tmp_path / "tax.blucache")
    assert cli.main(["cache-db", tj, cache] + (["-u"] if use_taxid else [])) == 0
    st_json, ck_json = pipeline.ingest_only(bt, tj, use_taxid)
    st_bin, ck_bin = pipeline.ingest_only(bt, cache, use_taxid)
    assert ck_json == ck_bin
    for k in ("n_hits", "n_queries", "n_taxids", "n_unmatched_rows"):
        assert st_json[k] == st_bin[k]
    assert st_json["n_taxids"] == 3000 and st_json["n_unmatched_rows"] > 0
    # wrong flavour, truncation and bit rot are refused
    with pytest.raises(N.BluError) as e:
        pipeline.ingest_only(bt, cache, not use_taxid)
    assert e.value.code == N.BLU_ERR_INVALID_ARG
    raw = open(cache, "rb").read()
    open(cache, "wb").write(raw[:-8])
    with pytest.raises(N.BluError, match="size mismatch"):
        pipeline.ingest_only(bt, cache, use_taxid)
    flipped = bytearray(raw)
    flipped[len(raw) // 2] ^= 0x40
    open(cache, "wb").write(bytes(flipped))
    with pytest.raises(N.BluError, match="checksum|inconsistent"):
        pipeline.ingest_only(bt, cache, use_taxid)
