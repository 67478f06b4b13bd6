"""Drop-in use-case end to end on the GPU: outfmt-6 TSV + blutils DB JSON + custom cutoffs file in, the
reference's result documents out (BASELINE config #1), compared field by field — consensus beans included —
with the string-faithful oracle fed from an independent (Python) reading of the same files."""
import json
import os

import numpy as np
import pytest

from blutils_amd import _native as N
from blutils_amd import pipeline, synth
from oracle import oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, tax, hits, use_taxid=False, scramble=True, half_scores=True):
    lineages_text = tax.lineage_strings(text=True)
    lineages_num = tax.lineage_strings(text=False)
    db = {"blutilsVersion": "8.3.1", "ignoreTaxids": None, "replaceRank": None, "dropNonLinnaeanTaxonomies": None,
          "sourceDatabase": "synthetic", "taxonomies": [
              {"taxid": int(tax.taxid[t]), "rank": "species", "numericLineage": lineages_num[t],
               "textLineage": lineages_text[t], "accessions": [{"accession": f"A{t}", "oid": str(t)}]} for t in range(tax.n)]}
    tj = tmp_path / "tax.blutils.json"
    tj.write_text(json.dumps(db))
    seg = hits["seg_off"]
    acc = hits["acc_rank"].view(np.uint32)
    rows = []
    rng = np.random.default_rng(9)
    for q in range(len(seg) - 1):
        s, e = int(seg[q]), int(seg[q + 1])
        top = hits["bitscore"][s:e].max() if e > s else 0
        for i in range(s, e):
            t = int(hits["tax_row"][i])
            taxid = int(tax.taxid[t]) if t >= 0 else 999999999
            bs = int(hits["bitscore"][i])
            # fractional scores truncate toward zero (mod.rs:184): x.5 on a non-top row stays below the top group
            bs_txt = f"{bs}.5" if (half_scores and bs < top - 1 and rng.random() < 0.2) else (f"{bs}" if rng.random() < 0.5 else f"{bs}.0")
            rows.append((q, f'q{q:08d}\tNR_{int(acc[i]):010d}.1\t{taxid}\t{hits["pident"][i]:.3f}\t{int(hits["align_len"][i])}'
                            f'\t3\t1\t1\t400\t5\t404\t1e-120\t{bs_txt}'))
    if scramble:   # rows of one query need not be contiguous; their relative (file) order must survive
        order = sorted(range(len(rows)), key=lambda i: (rng.integers(0, 4), i))
        rows = [rows[i] for i in order]
    bt = tmp_path / "blast.out.tsv"
    bt.write_text("\n".join(r[1] for r in rows) + "\n")
    return str(bt), str(tj), [r[0] for r in rows]


def _oracle_from_files(bt, tj, use_taxid, taxon, strategy, custom):
    """Independent reading of the two files -> faithful oracle."""
    db = json.load(open(tj))
    lin = {}                                           # taxid -> its lineages, one per listing (the left join of mod.rs:72-76
    for t in db["taxonomies"]:                         # gives one joined row per matching taxonomy row, in the DB's order)
        lin.setdefault(int(t["taxid"]), []).append(t["numericLineage"] if use_taxid else t["textLineage"])
    lineages = list(dict.fromkeys(s for v in lin.values() for s in v))
    lin_idx = {s: i for i, s in enumerate(lineages)}
    per_q = {}
    for line in open(bt):
        c = line.rstrip("\n").split("\t")
        per_q.setdefault(c[0], []).append(c)
    names = list(per_q)
    seg, acc_idx, accs, tax_row, pid, aln, bsc = [0], [], {}, [], [], [], []
    for qn in names:
        for c in per_q[qn]:
            for lineage in lin.get(int(c[2]), [None]):
                acc_idx.append(accs.setdefault(c[1], len(accs)))
                tax_row.append(lin_idx[lineage] if lineage is not None else -1)
                pid.append(float(c[3])); aln.append(int(c[4])); bsc.append(int(float(c[12])))
        seg.append(len(acc_idx))
    tab = orc.HitTable(np.array(seg, np.uint64), np.array(acc_idx, np.uint32), list(accs), np.array(tax_row, np.int64),
                       lineages, np.array(pid), np.array(aln, np.int64), np.array(bsc, np.int64))
    res = orc.run(tab, taxon=taxon, strategy=strategy, custom=custom, threads=4).results()
    return dict(zip(names, res))


@pytest.mark.parametrize("strategy,use_taxid,fmt", [("relaxed", False, "json"), ("cautious", True, "jsonl")])
def test_c1_files_through_the_pipeline(tmp_path, golden_dir, strategy, use_taxid, fmt):
    tax = synth.make_taxonomy(2000, synth.SEEDS["C1"])
    hits = synth.make_hits(tax, 1000, synth.SEEDS["C1"], 10, p_unmatched=0.002).numpy()
    bt, tj, _ = _write_inputs(tmp_path, tax, hits, use_taxid)
    vals = json.load(open(os.path.join(golden_dir, "custom_taxon_cutoffs_bacteria_16S.json")))["values"]
    cy = tmp_path / "custom-taxon-cutoffs-bacteria-16S.yaml"
    cy.write_text("".join(f"{k}: {v}\n" for k, v in vals.items()))
    custom = pipeline.custom_taxon_from_file(str(cy))
    headers = [f"q{q:08d}" for q in range(1000)] + ["fasta_only_1", "fasta_only_0"]
    # strict mode mirrors the reference: a query on which it panics fails the call
    with pytest.raises(N.BluError) as e:
        pipeline.build_consensus_identities(bt, tj, "custom", strategy, use_taxid, custom, headers=headers, out_format=fmt)
    assert e.value.code == pipeline.BLU_ERR_REFERENCE_PANIC
    got, stats = pipeline.build_consensus_identities(bt, tj, "custom", strategy, use_taxid, custom, headers=headers,
                                                     out_format=fmt, lenient=True)
    assert stats["n_queries"] == 1000 and stats["n_hits"] == 10000 and stats["n_taxids"] == 2000
    assert [g["query"] for g in got] == sorted(headers)                       # write_blutils_output.rs:111
    assert len({g["runId"] for g in got}) == 1
    exp = _oracle_from_files(bt, tj, use_taxid, "custom", strategy, custom)
    n_found = 0
    for g in got:
        if g["query"].startswith("fasta_only"):
            assert g["taxon"] is None                                          # mod.rs:86-102
            continue
        o = exp[g["query"]]
        if o["status"] != orc.ST_CONSENSUS:
            assert g["taxon"] is None, g["query"]
            continue
        n_found += 1
        assert g["taxon"] == o["taxon"], (g["query"], g["taxon"], o["taxon"])  # every field, beans and accession order included
    assert n_found > 900


def test_a_duplicated_taxid_gives_the_document_of_the_left_join(tmp_path):
    """mod.rs:72-76: a taxid the DB lists twice turns every hit of that subject into two joined rows — a single top hit
    becomes a two-row group, `occurrences` double.  Expected document: the faithful oracle fed the duplicated rows."""
    tax = synth.make_taxonomy(2000, synth.SEEDS["C1"])
    hits = synth.make_hits(tax, 1000, synth.SEEDS["C1"], 10, p_unmatched=0.002).numpy()
    bt, tj, _ = _write_inputs(tmp_path, tax, hits, False)
    db = json.load(open(tj))
    rng = np.random.default_rng(4)
    used = sorted({int(t) for t in hits["tax_row"] if t >= 0})
    for t in rng.choice(used, 60, replace=False):           # the same lineage again (a plain duplicate) or another taxon's
        e = dict(db["taxonomies"][int(t)])
        if rng.random() < 0.5:
            other = db["taxonomies"][int(rng.choice(used))]
            e["textLineage"], e["numericLineage"] = other["textLineage"], other["numericLineage"]
        db["taxonomies"].insert(int(rng.integers(0, len(db["taxonomies"]))), e)
    open(tj, "w").write(json.dumps(db))
    for strategy in ("relaxed", "cautious"):
        got, stats = pipeline.build_consensus_identities(bt, tj, "bacteria", strategy, lenient=True)
        assert stats["n_hits"] > 10000                       # the join multiplied rows
        exp = _oracle_from_files(bt, tj, False, "bacteria", strategy, None)
        n_found = 0
        for g in got:
            o = exp[g["query"]]
            if o["status"] != orc.ST_CONSENSUS:
                assert g["taxon"] is None, g["query"]
                continue
            n_found += 1
            assert g["taxon"] == o["taxon"], (g["query"], g["taxon"], o["taxon"])
        assert n_found > 900


def test_binary_taxonomy_cache_gives_the_same_document(tmp_path):
    """the cache of the taxonomies file is a drop-in for the JSON (same interning, same records, same text)."""
    tax = synth.make_taxonomy(2000, synth.SEEDS["C1"])
    hits = synth.make_hits(tax, 1000, synth.SEEDS["C1"], 10, p_unmatched=0.002).numpy()
    for use_taxid in (False, True):
        bt, tj, _ = _write_inputs(tmp_path, tax, hits, use_taxid)
        cache = str(tmp_path / f"tax.{int(use_taxid)}.blucache")
        pipeline.build_db_cache(tj, cache, use_taxid)
        a, _ = pipeline.build_consensus_identities(bt, tj, "bacteria", "relaxed", use_taxid, lenient=True, parse=False)
        b, st = pipeline.build_consensus_identities(bt, cache, "bacteria", "relaxed", use_taxid, lenient=True, parse=False)
        ja, jb = json.loads(a), json.loads(b)
        for r in ja["results"] + jb["results"]:
            r["runId"] = None                                                  # a fresh UUID per call
        assert ja == jb and st["n_taxids"] == 2000


def test_json_text_layout(tmp_path):
    """Byte layout of the JSON document = serde_json::to_string_pretty of BlutilsOutput{results, config: None}."""
    (tmp_path / "t.json").write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": 10, "rank": "species", "numericLineage": "d__2;g__5;s__10", "textLineage": "d__bacteria;g__ba;s__ba-x", "accessions": []},
        {"taxid": 11, "rank": "species", "numericLineage": "d__2;g__5;s__11", "textLineage": "d__bacteria;g__ba;s__ba-y", "accessions": []}]}))
    (tmp_path / "b.tsv").write_text(
        'q1\tACC_B.1\t11\t98.000\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n'
        '"q1"\tACC_A.1\t10\t99.500\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700.9\n'
        'q1\tACC_C.1\t10\t91.0\t380\t0\t0\t1\t400\t1\t400\t1e-40\t650\n')
    raw, _ = pipeline.build_consensus_identities(str(tmp_path / "b.tsv"), str(tmp_path / "t.json"), "bacteria", "relaxed",
                                                 parse=False)
    doc = json.loads(raw)
    run_id = doc["results"][0]["runId"]
    expected = json.dumps({"results": [{"runId": run_id, "query": "q1", "taxon": {
        "reachedRank": "genus", "maxAllowedRank": None, "identifier": "ba", "percIdentity": 99.5, "bitScore": 700.0,
        "taxonomy": "d__bacteria;g__ba", "mutated": False, "singleMatch": False, "consensusBeans": [
            {"rank": "species", "identifier": "ba-x", "occurrences": 1, "taxonomy": "d__bacteria;g__ba;s__ba-x", "accessions": ["ACC_A.1"]},
            {"rank": "species", "identifier": "ba-y", "occurrences": 1, "taxonomy": "d__bacteria;g__ba;s__ba-y", "accessions": ["ACC_B.1"]}]}}],
        "config": None}, indent=2)
    assert raw == expected


def test_more_than_three_decimals_keeps_the_f64_column(tmp_path):
    """The pipeline hands perc_identity over as milli-percent only when every value is exactly k/1000; a table with
    finer values keeps the f64 column, so a 4th-decimal difference still orders the hits (mod.rs:278-283)."""
    (tmp_path / "t.json").write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": 10, "rank": "species", "numericLineage": "d__2;g__5;s__10", "textLineage": "d__bacteria;g__ba;s__ba-x", "accessions": []}]}))
    (tmp_path / "b.tsv").write_text(
        'q1\tACC_A.1\t10\t99.1234\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n'
        'q1\tACC_B.1\t10\t99.1233\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n'
        'q2\tACC_B.1\t10\t98.5\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n')
    for strategy, want in (("relaxed", 99.1234), ("cautious", 99.1233)):
        got, _ = pipeline.build_consensus_identities(str(tmp_path / "b.tsv"), str(tmp_path / "t.json"), "bacteria", strategy)
        by = {g["query"]: g["taxon"] for g in got}
        assert by["q1"]["percIdentity"] == want and by["q1"]["reachedRank"] == "species"
        assert by["q2"]["percIdentity"] == 98.5 and by["q2"]["singleMatch"] is True


def test_cli_and_yaml(tmp_path, capsys):
    """`blu blastn build-consensus` arguments (commands.rs:105-143): stdout = compact JSON, file = pretty JSON with the
    extension forced; YAML carries the same values as JSON (numeric identifiers stay strings)."""
    from blutils_amd import cli
    tax = synth.make_taxonomy(300, 3)
    hits = synth.make_hits(tax, 120, 4, 8, p_unmatched=0.0).numpy()
    hits["pident"][:] = np.maximum(hits["pident"], 61.0)          # no panics: the CLI mirrors the strict reference
    bt, tj, _ = _write_inputs(tmp_path, tax, hits, use_taxid=True, scramble=False)
    # level-0 disagreements make the reference panic: drop those queries from the file for the strict CLI run
    lenient, _ = pipeline.build_consensus_identities(bt, tj, "bacteria", "relaxed", True, lenient=True)
    bad = {g["query"] for g in lenient if g["taxon"] is None}
    lines = [l for l in open(bt) if l.split("\t")[0] not in bad]
    open(bt, "w").writelines(lines)
    base = ["blastn", "build-consensus", bt, "-t", tj, "--taxon", "bacteria", "--strategy", "relaxed", "-u"]
    assert cli.main(base) == 0
    out = capsys.readouterr().out
    assert "\n" not in out.strip() and out.startswith('{"results":[{"runId":"')
    doc = json.loads(out)
    assert doc["config"] is None and len(doc["results"]) == 120 - len(bad)
    assert cli.main(base + ["--blutils-out-file", str(tmp_path / "res.txt")]) == 0
    pretty = json.load(open(tmp_path / "res.json"))                                   # extension forced to .json
    strip = lambda rs: [{k: v for k, v in r.items() if k != "runId"} for r in rs]
    assert strip(pretty["results"]) == strip(doc["results"])
    assert cli.main(base + ["--blutils-out-file", str(tmp_path / "res"), "--out-format", "yaml"]) == 0
    import yaml
    y = yaml.safe_load(open(tmp_path / "res.yaml"))
    assert y["config"] is None and strip(y["results"]) == strip(doc["results"])
    assert all(isinstance(r["taxon"]["identifier"], str) for r in y["results"])      # '1234' stays a string
    assert cli.main(base + ["--blutils-out-file", str(tmp_path / "res"), "--out-format", "jsonl"]) == 0
    jl = open(tmp_path / "res.jsonl").read().splitlines()
    assert jl[0] == "null" and strip([json.loads(l) for l in jl[1:]]) == strip(doc["results"])
    with pytest.raises(SystemExit):
        cli.main(["blastn", "build-consensus", bt, "-t", tj, "--taxon", "custom", "--strategy", "relaxed"])


def test_parallel_ingest_and_render_are_deterministic(tmp_path):
    """>= 4096 queries and > 1 MiB of text take the multi-threaded ingest and render paths: same document as
    the single-threaded run (runId aside)."""
    import re
    tax = synth.make_taxonomy(1500, 8)
    hits = synth.make_hits(tax, 6000, 9, 12, p_unmatched=0.001).numpy()
    bt, tj, _ = _write_inputs(tmp_path, tax, hits, scramble=True)
    assert os.path.getsize(bt) > (1 << 20)
    docs = []
    for threads in ("1", "7"):
        os.environ["BLU_INGEST_THREADS"] = threads
        try:
            raw, st = pipeline.build_consensus_identities(bt, tj, "bacteria", "relaxed", lenient=True, parse=False)
        finally:
            os.environ.pop("BLU_INGEST_THREADS", None)
        assert st["n_queries"] == 6000
        docs.append(re.sub(r'"runId": "[0-9a-f-]+"', '"runId": "x"', raw))
    assert docs[0] == docs[1]
    assert len(json.loads(docs[0])["results"]) == 6000


def test_golden_taxon_objects_leave_the_product_writer_byte_for_byte(tmp_path, golden_dir):
    """The reference's only real output (test/mock/output/zymo-mock/blutils.consensus.json, serde_json::to_string_pretty,
    write_blutils_output.rs:138) against the PRODUCT writer: the 253 distinct `taxon` objects are turned back into input
    FILES by the reconstruction recipe (tests/golden_recipe.py: one row per bean occurrence), the library builds the
    document from them, and every object that comes back with the same content must come back as the same BYTES as the
    reference wrote (key order, indentation, `845.0`, `null`, nested bean arrays).  Content equality itself is limited by
    the recipe (the beans keep one lineage per key), hence the floor on the count rather than 253."""
    import gzip
    with gzip.open(os.path.join(golden_dir, "zymo_mock_distilled.json.gz"), "rt") as f:
        cases = json.load(f)["cases"]
    with gzip.open(os.path.join(golden_dir, "zymo_mock_taxon_text.json.gz"), "rt") as f:
        text_of = json.load(f)["text_of"]
    lineages, rows = {}, []
    for i, c in enumerate(cases):
        t = c["taxon"]
        for bean in t["consensusBeans"]:
            taxid = lineages.setdefault(bean["taxonomy"], 1000 + len(lineages))
            for k in range(int(bean["occurrences"])):
                acc = bean["accessions"][min(k, len(bean["accessions"]) - 1)]
                # (align_length ascending in the bean's listed order: tests/golden_recipe.py — the golden's accession order is then the sort's)
                rows.append(f"case{i:04d}\t{acc}\t{taxid}\t{t['percIdentity']:.3f}\t{400 + k}\t0\t0\t1\t400\t1\t400\t1e-50\t{int(t['bitScore'])}")
    (tmp_path / "b.tsv").write_text("\n".join(rows) + "\n")
    (tmp_path / "t.json").write_text(json.dumps({"blutilsVersion": "7.1.3", "sourceDatabase": "golden", "taxonomies": [
        {"taxid": v, "rank": "", "numericLineage": k, "textLineage": k, "accessions": []} for k, v in lineages.items()]}))
    raw, _ = pipeline.build_consensus_identities(str(tmp_path / "b.tsv"), str(tmp_path / "t.json"), "bacteria", "relaxed", parse=False)
    doc = json.loads(raw)
    assert [r["query"] for r in doc["results"]] == [f"case{i:04d}" for i in range(len(cases))]

    def taxon_text(i):
        at = raw.index(f'"query": "case{i:04d}",\n      "taxon": ') + len(f'"query": "case{i:04d}",\n      "taxon": ')
        depth, j, in_str = 0, at, False
        while True:
            ch = raw[j]
            if in_str:
                if ch == "\\":
                    j += 1
                elif ch == '"':
                    in_str = False
            elif ch == '"':
                in_str = True
            elif ch == "{":
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0:
                    return raw[at:j + 1]
            j += 1

    same_content = same_bytes = 0
    for i, c in enumerate(cases):
        if doc["results"][i]["taxon"] == c["taxon"]:
            same_content += 1
            got, want = taxon_text(i), text_of[c["example_query"]]
            assert got == want, (c["example_query"], got[:300], want[:300])
            same_bytes += 1
    print(f"[golden writer] {same_content} of {len(cases)} objects reproduce in content, {same_bytes} of them byte for byte")
    assert same_bytes == same_content and same_bytes >= 100


def test_large_unsorted_table_streams_to_a_file_in_query_order(tmp_path):
    """70 000 queries whose names are not in file order: the parallel sort-and-merge of the result list, the writer thread
    that follows the renderers (out_path) and the replacement of an existing output file — the file must hold exactly
    the text the in-memory call returns, in byte order of the query names (write_blutils_output.rs:58-63, 111)."""
    import re
    nq, taxa = 70000, 3000
    tj = tmp_path / "t.json"
    tj.write_text(json.dumps({"blutilsVersion": "8.3.1", "sourceDatabase": "synthetic", "taxonomies": [
        {"taxid": 1000 + t, "rank": "species", "numericLineage": f"d__2;f__{t // 96};g__{t // 12};s__{1000 + t}",
         "textLineage": f"d__bacteria;f__fam{t // 96};g__gen{t // 12};s__sp{t}", "accessions": []} for t in range(taxa)]}))
    rng = np.random.default_rng(5)
    perm = rng.permutation(nq)
    sub = (np.repeat(rng.integers(0, taxa, nq), 3) + rng.integers(0, 12, 3 * nq)) % taxa
    pid = rng.integers(80000, 100001, 3 * nq)
    bs = np.repeat(rng.integers(200, 2000, nq), 3) - rng.integers(0, 2, 3 * nq)   # ties in about half of the top groups
    bt = tmp_path / "b.tsv"
    bt.write_text("".join(f"r{perm[i // 3]:07d}\tNR_{sub[i]:06d}.1\t{1000 + sub[i]}\t{pid[i] // 1000}.{pid[i] % 1000:03d}\t400\t3\t1\t1\t400\t5\t404\t1e-120\t{bs[i]}\n"
                          for i in range(3 * nq)))
    assert os.path.getsize(bt) > (1 << 20)
    raw, st = pipeline.build_consensus_identities(str(bt), str(tj), "bacteria", "relaxed", out_format="jsonl", lenient=True, parse=False)
    assert st["n_queries"] == nq and pipeline.last_ingest_path() == "gpu"
    lines = raw.splitlines()
    assert lines[0] == "null" and len(lines) == nq + 1
    names = [json.loads(l)["query"] for l in lines[1:]]
    assert names == sorted(f"r{k:07d}" for k in range(nq))
    strip = lambda s: re.sub(r'"runId":"[0-9a-f-]+"', '"runId":"x"', s)
    # the CPU ingest (host columns, top-score rows picked on the host, strings cut out of the mapped file) must give the
    # same document as the GPU ingest (columns left on the device, top-score rows compacted there, strings sent back packed)
    os.environ["BLU_INGEST"] = "cpu"
    try:
        raw_cpu, _ = pipeline.build_consensus_identities(str(bt), str(tj), "bacteria", "relaxed", out_format="jsonl", lenient=True, parse=False)
        assert pipeline.last_ingest_path() == "cpu"
    finally:
        os.environ.pop("BLU_INGEST", None)
    assert strip(raw_cpu) == strip(raw)
    # the fallback of the GPU path (the device could not hold the engine's work buffers): columns downloaded after the GPU
    # ingest, engine through the host-pointer staging path, top-score rows picked on the host
    os.environ["BLU_PIPELINE_HOST_COLUMNS"] = "1"
    try:
        raw_fb, _ = pipeline.build_consensus_identities(str(bt), str(tj), "bacteria", "relaxed", out_format="jsonl", lenient=True, parse=False)
        assert pipeline.last_ingest_path() == "gpu"
    finally:
        os.environ.pop("BLU_PIPELINE_HOST_COLUMNS", None)
    assert strip(raw_fb) == strip(raw)
    outp = tmp_path / "consensus.jsonl"
    outp.write_text("an older result\n" * 100000)
    for _ in range(2):   # the second run replaces the first run's file
        _, st2 = pipeline.build_consensus_identities(str(bt), str(tj), "bacteria", "relaxed", out_format="jsonl", lenient=True, parse=False,
                                                     out_path=str(outp))
        assert strip(outp.read_text()) == strip(raw)
