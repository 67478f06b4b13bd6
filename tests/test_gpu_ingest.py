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


def _rows(n_q, hits, rng, long_names=False):
    rows = []
    for q in range(n_q):
        name = f"query_with_a_rather_long_identifier_{q:07d}/1" if long_names else f"q{q:06d}"
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


@pytest.mark.parametrize("layout", ["grouped", "scrambled", "crlf_no_final_newline", "long_names"])
def test_gpu_ingest_gives_the_cpu_columns(tmp_path, force_gpu, layout):
    rng = np.random.default_rng(5)
    rows = _rows(6000, 12, rng, long_names=layout == "long_names")
    if layout == "scrambled":       # rows of one query need not be contiguous; their relative order must survive
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
    assert gs["n_hits"] == len(rows) and gs["n_queries"] == 6000 and gs["n_unmatched_rows"] > 0


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
