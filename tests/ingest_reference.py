"""Independent reading of an outfmt-6 table + blutils DB JSON, written with nothing but Python's own `str.split`,
`float()` and `int()`: the expected value of every SoA column the ingest produces, and of the checksum
`blu_ingest_only[_on]` reports over them.  Test infrastructure — shares no code with either product parser
(csrc/pipeline.cpp, csrc/ingest_gpu.hip).

What the columns are (reference: core/src/use_cases/build_consensus_identities/mod.rs):
  * 13 tab-separated fields, no header; schema query:str, subject_accession:str, subject_taxid:i64, perc_identity:f64,
    align_length:i64, ..., bit_score:f64 (mod.rs:226-244)
  * rows regrouped by query, queries in first-appearance order, file order kept inside a query (mod.rs:134-221)
  * bit_score truncated toward zero (mod.rs:184), quotes stripped from the two strings (mod.rs:169-176)
  * left join on the taxid (mod.rs:72-76): row of the DB's `taxonomies` list, 0xFFFFFFFF when absent; a taxid the DB lists m
    times gives m joined rows per hit (left-row order kept, the matches in the DB's order)
  * acc_rank = rank of the accession among the distinct accessions in byte order (String::cmp,
    find_multi_taxa_consensus.rs:59-66)
"""
import json

import numpy as np

UNMATCHED = 0xFFFFFFFF


def read_table(blast_path, db_path):
    db = json.load(open(db_path))
    rows_of = {}                                   # a taxid listed m times: m joined rows per hit, in the DB's order (left join)
    for i, t in enumerate(db["taxonomies"]):
        rows_of.setdefault(int(t["taxid"]), []).append(i)
    data = open(blast_path, "rb").read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    per_q = {}
    for ln in lines:
        if ln.endswith(b"\r"):
            ln = ln[:-1]
        c = ln.split(b"\t")
        assert len(c) >= 13, ln
        q = c[0].replace(b'"', b"")
        acc = c[1].replace(b'"', b"")
        taxid = int(c[2].decode())                 # Int64 column: no fraction, no exponent
        pid = float(c[3].decode())
        aln = int(c[4].decode())
        bs = int(float(c[12].decode()))            # truncation toward zero
        for trow in rows_of.get(taxid, [UNMATCHED]):
            per_q.setdefault(q, []).append((acc, trow, pid, aln, bs))
    accs = sorted({r[0] for rows in per_q.values() for r in rows})
    rank = {a: i for i, a in enumerate(accs)}
    seg = [0]
    cols = {"bitscore": [], "align_len": [], "tax_desc_row": [], "acc_rank": [], "pident": []}
    for q, rows in per_q.items():
        for acc, trow, pid, aln, bs in rows:
            cols["bitscore"].append(bs); cols["align_len"].append(aln); cols["tax_desc_row"].append(trow)
            cols["acc_rank"].append(rank[acc]); cols["pident"].append(pid)
        seg.append(len(cols["bitscore"]))
    return {
        "seg_off": np.array(seg, dtype=np.uint64),
        "bitscore": np.array(cols["bitscore"], dtype=np.int32),
        "align_len": np.array(cols["align_len"], dtype=np.int32),
        "tax_desc_row": np.array(cols["tax_desc_row"], dtype=np.uint32),
        "acc_rank": np.array(cols["acc_rank"], dtype=np.uint32),
        "pident": np.array(cols["pident"], dtype=np.float64),
        "query_names": list(per_q),
        "accessions": accs,
    }


def _fnv1a(h, b):
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def checksum(t) -> int:
    """The hash include/blu_pipeline.h documents for blu_ingest_only: FNV-1a over seg_off, bitscore, align_len, the joined
    taxonomy rows, acc_rank, pident, then every query name and every accession with its terminating NUL."""
    h = 1469598103934665603
    for k in ("seg_off", "bitscore", "align_len", "tax_desc_row", "acc_rank", "pident"):
        h = _fnv1a(h, t[k].tobytes())
    for s in t["query_names"]:
        h = _fnv1a(h, s + b"\0")
    for s in t["accessions"]:
        h = _fnv1a(h, s + b"\0")
    return h
