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
