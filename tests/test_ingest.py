"""Text ingest (outfmt-6 TSV + blutils DB JSON -> SoA), CPU only: the multi-threaded parse gives the same columns for
any thread count, malformed input is a call-level error, quoted / fractional fields follow mod.rs:169-184."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from blutils_amd import _native as N
from blutils_amd import pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(tmp_path, n_q=4000, hits=12, scramble=True):
    rng = np.random.default_rng(1)
    tj = tmp_path / "t.json"
    tj.write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": 100 + t, "rank": "species", "numericLineage": f"d__2;g__{t // 7};s__{100 + t}",
         "textLineage": f"d__b;g__g{t // 7};s__s{t}", "accessions": []} for t in range(500)]}))
    rows = []
    for q in range(n_q):
        for j in range(hits):
            taxid = 100 + int(rng.integers(0, 520))         # some taxids are not in the DB
            rows.append(f'"q{q:06d}"\tACC{int(rng.integers(0, 3000)):05d}.1\t{taxid}\t{80 + int(rng.integers(0, 20000)) / 1000:.3f}\t'
                        f'{380 + int(rng.integers(0, 100))}\t1\t0\t1\t400\t1\t400\t1e-50\t{500 + int(rng.integers(0, 200))}{".5" if j % 5 == 0 else ""}')
    if scramble:
        order = sorted(range(len(rows)), key=lambda i: (int(rng.integers(0, 3)), i))
        rows = [rows[i] for i in order]
    bt = tmp_path / "b.tsv"
    bt.write_text("\n".join(rows) + "\n")
    return str(bt), str(tj)


def _ingest_in_subprocess(bt, tj, threads):
    code = ("import sys, json; sys.path.insert(0, %r); from blutils_amd import pipeline; "
            "st, ck = pipeline.ingest_only(%r, %r); print(json.dumps([st, ck]))" % (ROOT, bt, tj))
    env = dict(os.environ, BLU_INGEST_THREADS=str(threads))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    return json.loads(out.strip().splitlines()[-1])


def test_ingest_is_thread_count_invariant(tmp_path):
    bt, tj = _files(tmp_path)                 # > 1 MiB, so the parallel path is taken
    assert os.path.getsize(bt) > (1 << 20)
    ref_stats, ref_ck = _ingest_in_subprocess(bt, tj, 1)
    assert ref_stats["n_hits"] == 48000 and ref_stats["n_queries"] == 4000 and ref_stats["n_taxids"] == 500
    assert 0 < ref_stats["n_unmatched_rows"] < 48000
    for threads in (2, 5, 8):
        st, ck = _ingest_in_subprocess(bt, tj, threads)
        assert ck == ref_ck and st["n_hits"] == ref_stats["n_hits"] and st["n_unmatched_rows"] == ref_stats["n_unmatched_rows"]


def test_ingest_errors(tmp_path):
    bt, tj = _files(tmp_path, n_q=10, hits=2, scramble=False)
    bad = tmp_path / "bad.tsv"
    bad.write_text("q1\tA.1\t100\t99.0\t400\n")
    with pytest.raises(N.BluError) as e:
        pipeline.ingest_only(str(bad), tj)
    assert e.value.code == N.BLU_ERR_PARSE if hasattr(N, "BLU_ERR_PARSE") else True
    with pytest.raises(N.BluError):
        pipeline.ingest_only(bt, str(tmp_path / "missing.json"))
    notnum = tmp_path / "notnum.tsv"
    notnum.write_text("q1\tA.1\tabc\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700\n")
    with pytest.raises(N.BluError):
        pipeline.ingest_only(str(notnum), tj)


# ---- the product parsers against an independent reading (tests/ingest_reference.py: str.split, float(), int()) ----------
def grammar_rows(gpu_forms_only: bool):
    """Rows whose numeric fields use every spelling the strict grammar admits.  gpu_forms_only: leave out the spellings the
    GPU parser hands to the CPU one (a leading '+', more than 15 significant digits, |exponent| beyond the one-operation
    fast path)."""
    pid_forms = ["99.356", "100", "100.000", "7e1", "9.9356e+01", "099.5", "97.", ".5e2", "8.05E1", "0.0", "-0.0", "66.667"]
    bs_forms = ["845", "845.0", "56.5", "7e2", "1E3", "0.5", "-0.0", "-3.7", "1.148e+05", "2147483647", "-2147483648.9", "1999.999"]
    tax_forms = ["105", "0105", "00000105", "-5", "999999999999", "100", "3099", "3100", "0", "-0", "101", "102"]
    aln_forms = ["400", "0400", "0", "-12", "2147483647", "-2147483648", "1", "38", "00038", "1500", "7", "9"]
    if not gpu_forms_only:
        pid_forms += ["+99.5", "99.123456789012345678", "1e-180", "1.7976931348623157e308", "0.000000000000000000001e23", "12345678901234567e-15"]
        bs_forms += ["+7", "7.00000000000000000001", "1e-180", "123456789.12345678", "1e9", "+0.0"]
        tax_forms += ["+105", "+0", "9223372036854775807", "-9223372036854775808", "000000000000000000000105", "+3099"]
        aln_forms += ["+400", "+0", "000000000000000000000400", "+1", "-0", "+2147483647"]
    rows = []
    n = len(pid_forms)
    for i in range(4 * n):
        q = f"query{i // 3:04d}"                              # three rows per query, rows of a query not always adjacent
        acc = ["NR_000105.1", "a_much_longer_accession_string_00000105.12", "NR_000105.10", "NR_000105", "B"][i % 5]
        dead = ["1\t0\t1\t400\t1\t400\t1e-50", "\t\t\t\t\t\t", "x\ty\tz\t-\t \tNaN\tinf"][i % 3]   # the dead columns are never parsed
        rows.append(f"{q}\t{acc}\t{tax_forms[i % n]}\t{pid_forms[(i * 5 + 1) % n]}\t{aln_forms[(i * 7 + 2) % n]}\t{dead}\t{bs_forms[(i * 11 + 3) % n]}")
    return rows


def grammar_db(tmp_path):
    tj = tmp_path / "g.json"
    tj.write_text(json.dumps({"blutilsVersion": "x", "sourceDatabase": "y", "taxonomies": [
        {"taxid": t, "rank": "species", "numericLineage": f"d__2;g__{t // 7};s__{t}",
         "textLineage": f"d__b;g__g{t // 7};s__s{t}", "accessions": []} for t in list(range(100, 3100)) + [0, 999999999999]]}))   # (taxid is a u64 in the DB: taxonomies_map.rs)
    return str(tj)


@pytest.mark.parametrize("eol", ["lf", "crlf", "crlf_no_final_newline"])
def test_cpu_parser_against_an_independent_reading(tmp_path, eol):
    from tests import ingest_reference as ref
    rows = grammar_rows(False)
    sep = "\n" if eol == "lf" else "\r\n"
    bt = tmp_path / "g.tsv"
    bt.write_bytes((sep.join(rows) + ("" if eol == "crlf_no_final_newline" else sep)).encode())
    tj = grammar_db(tmp_path)
    exp = ref.read_table(str(bt), tj)
    st, ck = pipeline.ingest_only(str(bt), tj, device=-1)
    assert st["n_hits"] == len(rows) and st["n_queries"] == len(exp["query_names"])
    assert st["n_unmatched_rows"] == int((exp["tax_desc_row"] == ref.UNMATCHED).sum()) > 0
    assert ck == ref.checksum(exp)


def test_int64_columns_take_no_fraction_and_floats_no_junk(tmp_path):
    """mod.rs:226-244: subject_taxid and align_length are Int64 (a fraction, an exponent or a blank fails the file as it
    fails the reference's typed CSV reader); a float field is a number from its first to its last byte."""
    tj = grammar_db(tmp_path)
    good = "q1\tA.1\t100\t99.0\t400\t0\t0\t1\t400\t1\t400\t1e-50\t700"
    for col, value in ((2, "100.0"), (2, "1e2"), (2, ""), (2, " 100"), (4, "400."), (4, "4e2"), (4, ""), (3, "99.0x"), (3, " 99.0"),
                       (3, ""), (3, "1_0"), (3, "0x10"), (12, "700 "), (12, ""), (12, "--7"), (12, "7e"), (12, "e7"), (12, ".")):
        c = good.split("\t")
        c[col] = value
        f = tmp_path / "strict.tsv"
        f.write_text(good + "\n" + "\t".join(c) + "\n")
        with pytest.raises(N.BluError, match="numeric"):
            pipeline.ingest_only(str(f), tj, device=-1)
