"""ctypes front end of the TEST-ONLY CPU oracle (oracle/blu_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (blutils_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLU_ORACLE_LIB: e.g. the -fsanitize=address,undefined build (make -C oracle asan) for a sanitizer pass on the CPU
_LIB_PATH = os.environ.get("BLU_ORACLE_LIB") or os.path.join(_HERE, "libblu_oracle.so")

TAXON = {"fungi": 0, "bacteria": 1, "eukaryotes": 2, "custom": 3}
STRATEGY = {"cautious": 0, "relaxed": 1}
# custom-cutoff field order (reference: core/src/domain/dtos/taxon.rs:16-25)
CUSTOM_FIELDS = ("domain", "kingdom", "phylum", "class", "order", "family", "genus", "species")

ST_CONSENSUS, ST_NO_CONSENSUS = 0, 1
ST_PANIC_PARSE, ST_PANIC_SINGLE_EMPTY, ST_PANIC_ROOT_DISAGREE = 2, 3, 4
ST_PANIC_INTERP_LEN, ST_PANIC_CUSTOM_MISSING, ST_PANIC_NAN_SORT, ST_PANIC_OTHER = 5, 6, 7, 8


class _Cfg(C.Structure):
    _fields_ = [
        ("taxon", C.c_int32),
        ("strategy", C.c_int32),
        ("has_custom", C.c_int32),
        ("custom", C.c_int16 * 8),
        ("custom_has", C.c_uint8 * 8),
        ("threads", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (make); returns the .so path."""
    srcs = [os.path.join(_HERE, f) for f in ("blu_oracle.cpp", "blu_oracle_columnar.cpp")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if os.environ.get("BLU_ORACLE_LIB"):
        return _LIB_PATH
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "libblu_oracle.so"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.blu_oracle_run.restype = C.c_void_p
        L.blu_oracle_run.argtypes = [
            C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Cfg),
        ]
        L.blu_oracle_status.restype = C.c_int32
        L.blu_oracle_status.argtypes = [C.c_void_p, C.c_uint64]
        L.blu_oracle_result_json.restype = C.c_void_p
        L.blu_oracle_result_json.argtypes = [C.c_void_p, C.c_uint64]
        L.blu_oracle_results_json.restype = C.c_void_p
        L.blu_oracle_results_json.argtypes = [C.c_void_p]
        L.blu_oracle_free_str.argtypes = [C.c_void_p]
        L.blu_oracle_free.argtypes = [C.c_void_p]
        L.blu_oracle_interpolate.restype = C.c_int32
        L.blu_oracle_interpolate.argtypes = [
            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
        ]
        L.blu_oracle_rank_display.restype = C.c_int32
        L.blu_oracle_rank_display.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        L.blu_oracle_rank_serde.restype = C.c_int32
        L.blu_oracle_rank_serde.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        _lib = L
    return _lib


def _custom_arrays(custom: Optional[dict]):
    vals = (C.c_int16 * 8)()
    has = (C.c_uint8 * 8)()
    if custom is not None:
        for i, k in enumerate(CUSTOM_FIELDS):
            if custom.get(k) is not None:
                vals[i] = int(custom[k])
                has[i] = 1
    return vals, has


def _make_cfg(taxon: str, strategy: str, custom: Optional[dict], threads: int) -> _Cfg:
    cfg = _Cfg()
    cfg.taxon = TAXON[taxon]
    cfg.strategy = STRATEGY[strategy]
    cfg.has_custom = 1 if custom is not None else 0
    vals, has = _custom_arrays(custom)
    for i in range(8):
        cfg.custom[i] = vals[i]
        cfg.custom_has[i] = has[i]
    cfg.threads = threads
    return cfg


class StringTable:
    """A char*[] kept alive on the Python side."""

    def __init__(self, strings: Sequence[str]):
        self._bytes = [s.encode("utf-8") for s in strings]
        self.array = (C.c_char_p * max(1, len(self._bytes)))(*self._bytes)
        self.n = len(self._bytes)

    @property
    def ptr(self):
        return C.cast(self.array, C.c_void_p)


@dataclass
class HitTable:
    """Rows grouped by query, file order inside a query (mod.rs:134-221)."""

    seg_off: np.ndarray          # uint64 [Q+1]
    acc_idx: np.ndarray          # uint32 [H] index into accessions
    accessions: Sequence[str]
    tax_row: np.ndarray          # int64 [H] index into lineages, -1 = taxid not in the DB
    lineages: Sequence[str]
    pident: np.ndarray           # float64 [H]
    align_len: np.ndarray        # int64 [H]
    bit_score: np.ndarray        # int64 [H] (already truncated, mod.rs:184)
    subject_taxid: Optional[np.ndarray] = None
    _acc_tab: Optional[StringTable] = field(default=None, repr=False)
    _lin_tab: Optional[StringTable] = field(default=None, repr=False)

    def tables(self):
        if self._acc_tab is None:
            self._acc_tab = StringTable(self.accessions)
        if self._lin_tab is None:
            self._lin_tab = StringTable(self.lineages)
        return self._acc_tab, self._lin_tab


class OracleRun:
    def __init__(self, handle, n):
        self._h = handle
        self.n = n

    def status(self, q: int) -> int:
        return lib().blu_oracle_status(self._h, q)

    def result(self, q: int) -> dict:
        p = lib().blu_oracle_result_json(self._h, q)
        try:
            return json.loads(C.string_at(p).decode("utf-8"))
        finally:
            lib().blu_oracle_free_str(p)

    def results(self) -> list:
        p = lib().blu_oracle_results_json(self._h)
        try:
            return json.loads(C.string_at(p).decode("utf-8"))
        finally:
            lib().blu_oracle_free_str(p)

    def close(self):
        if self._h:
            lib().blu_oracle_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run(table: HitTable, taxon: str = "bacteria", strategy: str = "relaxed",
        custom: Optional[dict] = None, threads: int = 1) -> OracleRun:
    """find_single_query_consensus over every query of `table` (mod.rs:104-128)."""
    seg = np.ascontiguousarray(table.seg_off, dtype=np.uint64)
    acc = np.ascontiguousarray(table.acc_idx, dtype=np.uint32)
    tax = np.ascontiguousarray(table.tax_row, dtype=np.int64)
    pid = np.ascontiguousarray(table.pident, dtype=np.float64)
    aln = np.ascontiguousarray(table.align_len, dtype=np.int64)
    bsc = np.ascontiguousarray(table.bit_score, dtype=np.int64)
    nq = len(seg) - 1
    nh = int(seg[-1]) if nq >= 0 and len(seg) else 0
    assert len(acc) == nh and len(tax) == nh and len(pid) == nh and len(aln) == nh and len(bsc) == nh
    acc_tab, lin_tab = table.tables()
    if nh:
        assert int(acc.max()) < max(1, acc_tab.n)
        assert int(tax.max()) < max(1, lin_tab.n)
    cfg = _make_cfg(taxon, strategy, custom, threads)
    stx = None
    if table.subject_taxid is not None:
        stx = np.ascontiguousarray(table.subject_taxid, dtype=np.int64)
    h = lib().blu_oracle_run(
        nq, seg.ctypes.data, acc.ctypes.data, acc_tab.ptr, tax.ctypes.data, lin_tab.ptr,
        stx.ctypes.data if stx is not None else None,
        pid.ctypes.data, aln.ctypes.data, bsc.ctypes.data, C.byref(cfg),
    )
    return OracleRun(h, nq)


def interpolate(ranks: Sequence[str], taxon: str = "bacteria", custom: Optional[dict] = None):
    """Cutoffs of one lineage rank sequence (linnaean_ranks.rs:220-383)."""
    tab = StringTable(list(ranks))
    out = np.zeros(len(ranks), dtype=np.float64)
    isdef = np.zeros(len(ranks), dtype=np.uint8)
    vals, has = _custom_arrays(custom)
    rc = lib().blu_oracle_interpolate(
        TAXON[taxon], 1 if custom is not None else 0, C.cast(vals, C.c_void_p), C.cast(has, C.c_void_p),
        len(ranks), tab.ptr, out.ctypes.data, isdef.ctypes.data,
    )
    if rc != 0:
        raise RuntimeError(f"oracle panic status {rc}")
    return out, isdef.astype(bool)


def rank_display(rank: str) -> str:
    buf = C.create_string_buffer(256)
    lib().blu_oracle_rank_display(rank.encode(), buf, 256)
    return buf.value.decode()


def rank_serde(rank: str) -> str:
    buf = C.create_string_buffer(256)
    lib().blu_oracle_rank_serde(rank.encode(), buf, 256)
    return buf.value.decode()


# ---------------------------------------------------------------------------
# Columnar oracle (oracle/blu_oracle_columnar.cpp): same semantics on the
# interned SoA layout; records have the layout of include/blu_consensus.h.
# ---------------------------------------------------------------------------
RESULT_DTYPE = np.dtype([
    ("status", "u1"), ("flags", "u1"), ("bean_index", "u1"), ("max_allowed_level", "u1"),
    ("reached_rank", "<u2"), ("max_allowed_rank", "<u2"), ("identifier_node", "<u4"), ("ref_row", "<u4"),
    ("level_mask", "<u8"), ("ident_used", "<f8"),
])
assert RESULT_DTYPE.itemsize == 32


def columnar_run(lin_off, lin_node, lin_rank, rank_names, seg_off, bitscore, tax_row, pident, align_len, acc_rank,
                 taxon="bacteria", strategy="relaxed", custom=None, bad=None, threads=1):
    L = lib()
    fn = L.blu_oracle_columnar_run
    fn.restype = C.c_int32
    fn.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lin_off = np.ascontiguousarray(lin_off, dtype=np.uint64)
    lin_node = np.ascontiguousarray(lin_node, dtype=np.uint32)
    lin_rank = np.ascontiguousarray(lin_rank, dtype=np.uint16)
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    bs = np.ascontiguousarray(bitscore, dtype=np.int32)
    tx = np.ascontiguousarray(tax_row).view(np.uint32) if np.asarray(tax_row).dtype == np.int32 else np.ascontiguousarray(tax_row, dtype=np.uint32)
    pid = np.ascontiguousarray(pident, dtype=np.float64)
    aln = np.ascontiguousarray(align_len, dtype=np.int32)
    ac = np.ascontiguousarray(acc_rank).view(np.uint32) if np.asarray(acc_rank).dtype == np.int32 else np.ascontiguousarray(acc_rank, dtype=np.uint32)
    names = StringTable(list(rank_names))
    vals, has = _custom_arrays(custom)
    badp = None
    if bad is not None:
        bad = np.ascontiguousarray(bad, dtype=np.uint8)
        badp = bad.ctypes.data
    nq = len(seg) - 1
    out = np.zeros(nq, dtype=RESULT_DTYPE)
    rc = fn(len(lin_off) - 1, lin_off.ctypes.data, lin_node.ctypes.data, lin_rank.ctypes.data, len(rank_names),
            names.ptr, badp, TAXON[taxon], 1 if custom is not None else 0, C.cast(vals, C.c_void_p),
            C.cast(has, C.c_void_p), nq, seg.ctypes.data, bs.ctypes.data, tx.ctypes.data, pid.ctypes.data,
            aln.ctypes.data, ac.ctypes.data, STRATEGY[strategy], threads, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle panic status {rc}")
    return out


# ---------------------------------------------------------------------------
# bench.py cpu_baseline leg: the string-faithful oracle on a slice of a synthetic
# SoA table.  String tables are built in C (data preparation, not timed).
# ---------------------------------------------------------------------------
def faithful_on_synthetic(lin_off, lin_node, lin_rank, rank_names, seg_off, bitscore, tax_row, pident, align_len,
                          acc_rank, taxon="bacteria", strategy="relaxed", custom=None, threads=1, want_json=False):
    """Returns (seconds spent inside blu_oracle_run, OracleRun).  Every row gets its own accession entry
    (acc_idx = row), so no dictionary pass is needed."""
    import time

    L = lib()
    L.blu_oracle_lineage_strings.restype = C.c_void_p
    L.blu_oracle_lineage_strings.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    L.blu_oracle_accession_strings.restype = C.c_void_p
    L.blu_oracle_accession_strings.argtypes = [C.c_uint64, C.c_void_p]
    L.blu_oracle_strtab_ptr.restype = C.c_void_p
    L.blu_oracle_strtab_ptr.argtypes = [C.c_void_p]
    L.blu_oracle_strtab_free.argtypes = [C.c_void_p]
    lin_off = np.ascontiguousarray(lin_off, dtype=np.uint64)
    lin_node = np.ascontiguousarray(lin_node, dtype=np.uint32)
    lin_rank = np.ascontiguousarray(lin_rank, dtype=np.uint16)
    names = StringTable(list(rank_names))
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    nq, nh = len(seg) - 1, int(seg[-1])
    acc = np.ascontiguousarray(acc_rank)
    acc = acc.view(np.uint32) if acc.dtype == np.int32 else np.ascontiguousarray(acc, dtype=np.uint32)
    tr = np.ascontiguousarray(tax_row)
    tr = tr.astype(np.int64) if tr.dtype == np.int32 else np.where(tr == 0xFFFFFFFF, -1, tr.astype(np.int64))
    tr = np.ascontiguousarray(tr, dtype=np.int64)
    pid = np.ascontiguousarray(pident, dtype=np.float64)
    aln = np.ascontiguousarray(align_len, dtype=np.int64)
    bsc = np.ascontiguousarray(bitscore, dtype=np.int64)
    acc_idx = np.arange(nh, dtype=np.uint32)
    lt = L.blu_oracle_lineage_strings(len(lin_off) - 1, lin_off.ctypes.data, lin_node.ctypes.data,
                                      lin_rank.ctypes.data, names.ptr, b"n")
    at = L.blu_oracle_accession_strings(nh, acc.ctypes.data)
    try:
        cfg = _make_cfg(taxon, strategy, custom, threads)
        t0 = time.perf_counter()
        h = L.blu_oracle_run(nq, seg.ctypes.data, acc_idx.ctypes.data, L.blu_oracle_strtab_ptr(at), tr.ctypes.data,
                             L.blu_oracle_strtab_ptr(lt), None, pid.ctypes.data, aln.ctypes.data, bsc.ctypes.data,
                             C.byref(cfg))
        dt = time.perf_counter() - t0
        run_ = OracleRun(h, nq)
        if want_json:
            run_._json = run_.results()
    finally:
        L.blu_oracle_strtab_free(lt)
        L.blu_oracle_strtab_free(at)
    return dt, run_
