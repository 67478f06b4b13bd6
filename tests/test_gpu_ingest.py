"""GPU ingest (csrc/ingest_gpu.hip) against the CPU ingest: identical SoA columns, query order and accession ranks
(one checksum over all of them) for plain BLAST tables; files outside the GPU parser's form take the CPU path."""
import json
import os

import numpy as np
import pytest

from blutils_amd import _native as N
from blutils_amd import pipeline

pytestmark = pytest.mark.gpu


@pytest.fixture()
def force_gpu():
    old = os.environ.get("BLU_INGEST")
    os.environ["BLU_INGEST"] = "gpu"
    yield
    if old is None:
        os.environ.pop("BLU_INGEST", None)
    else:
        os.environ["BLU_INGEST"] = old


def _db(tmp_path, n=3000):
    tj = tmp_path / "t.json"
    tj.write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": 100 + t, "rank": "species", "numericLineage": f"d__2;g__{t // 7};s__{100 + t}",
         "textLineage": f"d__b;g__g{t // 7};s__s{t}", "accessions": []} for t in range(n)]}))
    return str(tj)


def _rows(n_q, hits, rng, long_names=False, long_lines=False):
    rows = []
    for q in range(n_q):
        name = f"query_with_a_rather_long_identifier_{q:07d}/1" if long_names else f"q{q:06d}"
        if long_lines and (q // 500) % 2:      # stretches of 300-byte lines: their 256-line blocks do not fit the parse kernel's
            name = f"q{q:06d}_" + "x" * 260    # LDS stage and take its general form; the accessions are shared with the others
        for j in range(int(rng.integers(1, hits + 1))):
            t = int(rng.integers(0, 3100))                       # some taxids are not in the DB
            bs = int(rng.integers(50, 200000))
            bs_txt = f"{bs / 1000:.3f}e+03" if bs >= 99999 else (f"{bs}.5" if j % 5 == 0 else str(bs))   # BLAST prints large scores as 1.148e+05
            acc = f"NR_{t:06d}.1" if t % 3 else f"a_much_longer_accession_string_{t:08d}.12"
            rows.append(f"{name}\t{acc}\t{100 + t}\t{80 + int(rng.integers(0, 20001)) / 1000:.3f}\t{int(rng.integers(100, 2000))}"
                        f"\t1\t0\t1\t400\t1\t400\t{10.0 ** -int(rng.integers(3, 180)):.2e}\t{bs_txt}")
    return rows


def _both(bt, tj):
    cpu_stats, cpu_ck = pipeline.ingest_only(bt, tj, False, device=-1)
    assert pipeline.last_ingest_path() == "cpu"
    gpu_stats, gpu_ck = pipeline.ingest_only(bt, tj, False, device=0)
    return cpu_stats, cpu_ck, gpu_stats, gpu_ck, pipeline.last_ingest_path()


@pytest.mark.parametrize("layout", ["grouped", "scrambled", "scrambled_large", "crlf_no_final_newline", "long_names", "long_lines", "extra_columns"])
def test_gpu_ingest_gives_the_cpu_columns(tmp_path, force_gpu, layout):
    rng = np.random.default_rng(5)
    n_q = 70000 if layout == "scrambled_large" else 6000       # (70 000 ids: three 8-bit passes of the regrouping sort, 60 blocks)
    rows = _rows(n_q, 6 if layout == "scrambled_large" else 12, rng, long_names=layout == "long_names",
                 long_lines=layout in ("long_lines", "extra_columns"))
    if layout == "extra_columns":            # a 14th and 15th column are ignored (both forms of the parse kernel: the long stretches too)
        rows = [r + "\textra\t1.5" if i % 3 == 0 else r for i, r in enumerate(rows)]
    if layout.startswith("scrambled"):       # rows of one query need not be contiguous; their relative order must survive
        order = sorted(range(len(rows)), key=lambda i: (int(rng.integers(0, 4)), i))
        rows = [rows[i] for i in order]
    text = "\n".join(rows) + "\n"
    if layout == "crlf_no_final_newline":
        text = "\r\n".join(rows)
    bt = tmp_path / "b.tsv"
    bt.write_bytes(text.encode())
    cs, cck, gs, gck, path = _both(str(bt), _db(tmp_path))
    assert path == "gpu"
    assert gck == cck
    for k in ("n_hits", "n_queries", "n_unmatched_rows"):
        assert cs[k] == gs[k]
    assert gs["n_hits"] == len(rows) and gs["n_queries"] == n_q and gs["n_unmatched_rows"] > 0


@pytest.fixture(scope="module")
def big_table(tmp_path_factory):
    rng = np.random.default_rng(21)
    rows = _rows(40000, 20, rng, long_names=True)
    d = tmp_path_factory.mktemp("big")
    bt = d / "b.tsv"
    bt.write_bytes(("\n".join(rows) + "\n").encode())
    assert bt.stat().st_size > 5 * (8 << 20)
    cs, cck = pipeline.ingest_only(str(bt), _db(d), False, device=-1)
    return str(bt), _db(d), len(rows), cs, cck


@pytest.mark.parametrize("threads", ["1", "3", "5"])
def test_a_table_uploaded_in_many_pieces(big_table, force_gpu, threads):
    """50 MB of text: six 8 MiB pieces through the pinned staging slots (each reader thread reuses its two slots, so a slot's
    event is waited for), with one, three and five reader threads; the columns are the CPU parser's."""
    bt, tj, n_rows, cs, cck = big_table
    old = os.environ.get("BLU_UPLOAD_THREADS")
    os.environ["BLU_UPLOAD_THREADS"] = threads
    try:
        gs, gck = pipeline.ingest_only(bt, tj, False, device=0)
    finally:
        if old is None:
            os.environ.pop("BLU_UPLOAD_THREADS", None)
        else:
            os.environ["BLU_UPLOAD_THREADS"] = old
    assert pipeline.last_ingest_path() == "gpu" and gck == cck
    assert gs["n_hits"] == cs["n_hits"] == n_rows and gs["n_queries"] == 40000


@pytest.mark.parametrize("case", ["quoted", "empty_line", "many_digits"])
def test_files_outside_the_gpu_form_take_the_cpu_path(tmp_path, force_gpu, case):
    rng = np.random.default_rng(6)
    rows = _rows(300, 5, rng)
    if case == "quoted":
        rows[17] = '"' + rows[17].replace("\t", '"\t', 1)
    elif case == "empty_line":
        rows.insert(40, "")
    elif case == "many_digits":
        c = rows[9].split("\t"); c[3] = "99.12345678901234567"; rows[9] = "\t".join(c)
    bt = tmp_path / "b.tsv"
    bt.write_text("\n".join(rows) + "\n")
    cs, cck, gs, gck, path = _both(str(bt), _db(tmp_path))
    assert path == "cpu" and gck == cck and gs["n_hits"] == cs["n_hits"]


def test_errors_are_the_cpu_parsers(tmp_path, force_gpu):
    tj = _db(tmp_path)
    bad = tmp_path / "bad.tsv"
    bad.write_text("q1\tA.1\t100\t99.0\t400\n")
    with pytest.raises(N.BluError, match="columns"):
        pipeline.ingest_only(str(bad), tj, False, device=0)
    na = tmp_path / "na.tsv"
    na.write_text("q1\tA.1\tN/A\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n")
    with pytest.raises(N.BluError, match="numeric"):
        pipeline.ingest_only(str(na), tj, False, device=0)
    # a typed CSV column takes no blanks, an Int64 column no fraction: the GPU parser declines, the CPU parser refuses
    good = "q1\tA.1\t100\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n"
    for col, value in ((3, " 99.0"), (4, "400.0"), (2, "1e2"), (12, "700x")):
        c = good.rstrip("\n").split("\t")
        c[col] = value
        f = tmp_path / "strict.tsv"
        f.write_text(good + "\t".join(c) + "\n")
        with pytest.raises(N.BluError, match="numeric"):
            pipeline.ingest_only(str(f), tj, False, device=0)


def test_pipeline_document_is_the_same_with_either_parser(tmp_path, force_gpu):
    rng = np.random.default_rng(8)
    bt = tmp_path / "b.tsv"
    bt.write_text("\n".join(_rows(2000, 10, rng)) + "\n")
    tj = _db(tmp_path)
    a, _ = pipeline.build_consensus_identities(str(bt), tj, "bacteria", "relaxed", lenient=True)
    assert pipeline.last_ingest_path() == "gpu"
    os.environ["BLU_INGEST"] = "cpu"
    b, _ = pipeline.build_consensus_identities(str(bt), tj, "bacteria", "relaxed", lenient=True)
    assert pipeline.last_ingest_path() == "cpu"
    for r in a + b:
        r["runId"] = None
    assert a == b and sum(r["taxon"] is not None for r in a) > 1000


def test_garbage_input_is_refused_like_on_the_cpu(tmp_path, force_gpu):
    """Arbitrary bytes must not fault the GPU parser: it declines and the CPU parser reports its usual error."""
    rng = np.random.default_rng(9)
    tj = _db(tmp_path)
    good = "\n".join(_rows(200, 4, rng)) + "\n"
    variants = {
        "random_bytes": bytes(rng.integers(0, 256, 200000, dtype=np.uint8)),
        "no_newline_at_all": b"x" * 100000,
        "only_newlines": b"\n" * 5000,
        "truncated_mid_line": good.encode()[: len(good) // 2 - 7],
        "nul_bytes": good.replace("\t400\t", "\t4\x000\t", 3).encode(),
        "huge_field": (good + "q\t" + "A" * (1 << 21) + "\t100\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-5\t50\n").encode(),
    }
    for name, blob in variants.items():
        f = tmp_path / f"{name}.tsv"
        f.write_bytes(blob)
        outcome = []
        for dev in (-1, 0):
            try:
                st, ck = pipeline.ingest_only(str(f), tj, False, device=dev)
                outcome.append(("ok", st["n_hits"], ck))
            except N.BluError as e:
                outcome.append(("error", e.code))
        assert outcome[0] == outcome[1], (name, outcome)


# ---- the GPU parser against oracles that share nothing with either product parser ----------------------------------------
def _assert_columns_equal(got, exp):
    for k in ("seg_off", "bitscore", "align_len", "tax_desc_row", "acc_rank"):
        assert np.array_equal(got[k], exp[k]), k
    assert np.array_equal(got["pident"].view(np.uint64), exp["pident"].view(np.uint64))      # bit for bit
    assert got["query_names"] == exp["query_names"] and got["accessions"] == exp["accessions"]


@pytest.mark.parametrize("form", ["staged", "general"])
@pytest.mark.parametrize("eol", ["lf", "crlf", "crlf_no_final_newline"])
def test_gpu_parser_against_an_independent_reading(tmp_path, force_gpu, eol, form):
    """Every numeric spelling the GPU parser takes (tests/test_ingest.py: grammar_rows), columns compared one by one with
    tests/ingest_reference.py — Python's float() / int() on str.split fields, the reference's schema (mod.rs:226-244).
    form: the parse kernel's LDS-staged form (BLAST's usual line lengths) and its general form (256 lines that do not fit
    the stage: here a dead column carries 200 bytes of filler)."""
    from tests import ingest_reference as ref
    from tests.test_ingest import grammar_db, grammar_rows
    rows = grammar_rows(True) * 40                    # 1920 rows: several 256-line blocks of the parse kernel
    if form == "general":
        rows = ["\t".join(c[:5] + [c[5] + "x" * 200] + c[6:]) for c in (r.split("\t") for r in rows)]
    sep = "\n" if eol == "lf" else "\r\n"
    bt = tmp_path / "g.tsv"
    bt.write_bytes((sep.join(rows) + ("" if eol == "crlf_no_final_newline" else sep)).encode())
    tj = grammar_db(tmp_path)
    exp = ref.read_table(str(bt), tj)
    got = pipeline.ingest_columns(str(bt), tj, device=0)
    assert pipeline.last_ingest_path() == "gpu"
    _assert_columns_equal(got, exp)
    st, ck = pipeline.ingest_only(str(bt), tj, False, device=0)
    assert pipeline.last_ingest_path() == "gpu" and ck == ref.checksum(exp)
    assert st["n_unmatched_rows"] == int((exp["tax_desc_row"] == ref.UNMATCHED).sum()) > 0


def test_spellings_the_gpu_parser_declines_still_give_the_independent_columns(tmp_path, force_gpu):
    """A leading '+', 17-digit mantissas, 1e-180: the GPU parser hands the file over, and what comes back is still what
    Python reads."""
    from tests import ingest_reference as ref
    from tests.test_ingest import grammar_db, grammar_rows
    bt = tmp_path / "g.tsv"
    bt.write_text("\n".join(grammar_rows(False) * 10) + "\n")
    tj = grammar_db(tmp_path)
    got = pipeline.ingest_columns(str(bt), tj, device=0)
    assert pipeline.last_ingest_path() == "cpu"
    _assert_columns_equal(got, ref.read_table(str(bt), tj))


@pytest.mark.parametrize("layout", ["scrambled", "long_names", "long_lines"])
def test_gpu_parser_columns_of_blast_shaped_tables(tmp_path, force_gpu, layout):
    from tests import ingest_reference as ref
    rng = np.random.default_rng(15)
    rows = _rows(3000, 9, rng, long_names=layout == "long_names", long_lines=layout == "long_lines")
    if layout == "scrambled":
        order = sorted(range(len(rows)), key=lambda i: (int(rng.integers(0, 5)), i))
        rows = [rows[i] for i in order]
    bt = tmp_path / "b.tsv"
    bt.write_text("\n".join(rows) + "\n")
    tj = _db(tmp_path)
    got = pipeline.ingest_columns(str(bt), tj, device=0)
    assert pipeline.last_ingest_path() == "gpu"
    _assert_columns_equal(got, ref.read_table(str(bt), tj))


@pytest.mark.parametrize("strategy,use_taxid", [("relaxed", False), ("cautious", True)])
def test_gpu_parsed_c1_files_give_the_faithful_oracles_document(tmp_path, golden_dir, force_gpu, strategy, use_taxid):
    """BASELINE config 1 as files, the table parsed by the GPU (forced: it is under 1 MiB), every rendered field compared
    with the string-faithful oracle fed from an independent Python reading of the same two files."""
    from blutils_amd import synth
    from oracle import oracle as orc
    from tests.test_gpu_pipeline import _oracle_from_files, _write_inputs
    tax = synth.make_taxonomy(2000, synth.SEEDS["C1"])
    hits = synth.make_hits(tax, 1000, synth.SEEDS["C1"], 10, p_unmatched=0.002).numpy()
    bt, tj, _ = _write_inputs(tmp_path, tax, hits, use_taxid)
    vals = json.load(open(os.path.join(golden_dir, "custom_taxon_cutoffs_bacteria_16S.json")))["values"]
    custom = {k: v for k, v in vals.items() if v is not None}
    got, stats = pipeline.build_consensus_identities(bt, tj, "custom", strategy, use_taxid, custom, lenient=True)
    assert pipeline.last_ingest_path() == "gpu"
    assert stats["n_queries"] == 1000 and stats["n_hits"] == 10000
    exp = _oracle_from_files(bt, tj, use_taxid, "custom", strategy, custom)
    n_found = 0
    for g in got:
        o = exp[g["query"]]
        if o["status"] != orc.ST_CONSENSUS:
            assert g["taxon"] is None, g["query"]
            continue
        n_found += 1
        assert g["taxon"] == o["taxon"], (g["query"], g["taxon"], o["taxon"])
    assert n_found > 900
