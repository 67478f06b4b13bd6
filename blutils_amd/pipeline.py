"""ctypes front end of the host pipeline (include/blu_pipeline.h): the drop-in for the reference use-case

    build_consensus_identities(blast_output, taxonomies_file, taxon, strategy, use_taxid, custom_taxon_values)
    (core/src/use_cases/build_consensus_identities/mod.rs:40-47)  +  write_blutils_output (write_blutils_output.rs:33)

Same argument names and meaning; the consensus itself runs on the GPU (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Optional, Sequence

from . import _native as N

OUT_FORMAT = {"json": 0, "jsonl": 1, "yaml": 2, "json-compact": 3}
BLU_ERR_REFERENCE_PANIC = 9


class PipelineParams(C.Structure):
    _fields_ = [("cutoffs", N.CutoffConfig), ("strategy", C.c_int32), ("use_taxid", C.c_int32), ("device", C.c_int32),
                ("out_format", C.c_int32), ("lenient", C.c_int32), ("reserved", C.c_int32)]


class PipelineStats(C.Structure):
    _fields_ = [("n_hits", C.c_uint64), ("n_queries", C.c_uint64), ("n_taxids", C.c_uint64),
                ("n_unmatched_rows", C.c_uint64), ("t_load_db_s", C.c_double), ("t_load_hits_s", C.c_double),
                ("t_engine_s", C.c_double), ("t_render_s", C.c_double)]


def _bind():
    L = N.lib()
    L.blu_build_consensus_identities.restype = C.c_int
    L.blu_build_consensus_identities.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.POINTER(PipelineParams),
                                                 C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(PipelineStats)]
    L.blu_free_text.argtypes = [C.c_void_p]
    L.blu_custom_taxon_from_file.restype = C.c_int
    L.blu_custom_taxon_from_file.argtypes = [C.c_char_p, C.POINTER(N.CutoffConfig)]
    return L


def ingest_only(blast_output: str, taxonomies_file: str, use_taxid: bool = False, device: int = -1):
    """Text ingest alone: returns (stats, checksum of the SoA columns).  device < 0: CPU ingest (no GPU needed);
    device >= 0: the GPU parser where it applies (same columns, same checksum)."""
    L = _bind()
    L.blu_ingest_only_on.restype = C.c_int
    L.blu_ingest_only_on.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(PipelineStats), C.POINTER(C.c_uint64)]
    st, ck = PipelineStats(), C.c_uint64()
    rc = L.blu_ingest_only_on(blast_output.encode(), taxonomies_file.encode(), 1 if use_taxid else 0, device, C.byref(st), C.byref(ck))
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_ingest_only")
    return {f: getattr(st, f) for f, _ in PipelineStats._fields_}, ck.value


class IngestColumns(C.Structure):
    _fields_ = [("n_hits", C.c_uint64), ("n_queries", C.c_uint64), ("n_accessions", C.c_uint64), ("seg_off", C.POINTER(C.c_uint64)),
                ("bitscore", C.POINTER(C.c_int32)), ("align_len", C.POINTER(C.c_int32)), ("tax_desc_row", C.POINTER(C.c_uint32)),
                ("acc_rank", C.POINTER(C.c_uint32)), ("pident", C.POINTER(C.c_double)), ("query_names", C.c_void_p),
                ("query_names_bytes", C.c_uint64), ("accessions", C.c_void_p), ("accessions_bytes", C.c_uint64)]


def ingest_columns(blast_output: str, taxonomies_file: str, use_taxid: bool = False, device: int = -1) -> dict:
    """The SoA columns of the ingest (include/blu_pipeline.h: blu_ingest_columns_on) as numpy arrays + the two string tables."""
    import numpy as np
    L = _bind()
    L.blu_ingest_columns_on.restype = C.c_int
    L.blu_ingest_columns_on.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(IngestColumns)]
    L.blu_ingest_columns_free.argtypes = [C.POINTER(IngestColumns)]
    c = IngestColumns()
    rc = L.blu_ingest_columns_on(blast_output.encode(), taxonomies_file.encode(), 1 if use_taxid else 0, device, C.byref(c))
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_ingest_columns_on")
    try:
        nh, nq = int(c.n_hits), int(c.n_queries)
        arr = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt, copy=True)
        out = {"seg_off": arr(c.seg_off, nq + 1, np.uint64), "bitscore": arr(c.bitscore, nh, np.int32),
               "align_len": arr(c.align_len, nh, np.int32), "tax_desc_row": arr(c.tax_desc_row, nh, np.uint32),
               "acc_rank": arr(c.acc_rank, nh, np.uint32), "pident": arr(c.pident, nh, np.float64)}
        split = lambda p, n: C.string_at(p, n).split(b"\0")[:-1] if n else []
        out["query_names"] = split(c.query_names, int(c.query_names_bytes))
        out["accessions"] = split(c.accessions, int(c.accessions_bytes))
        return out
    finally:
        L.blu_ingest_columns_free(C.byref(c))


def last_ingest_path() -> str:
    """'gpu' or 'cpu': the parser the last ingest of this thread used."""
    return "gpu" if N.lib().blu_last_ingest_path() == 1 else "cpu"


def build_db_cache(taxonomies_file: str, cache_file: str, use_taxid: bool = False) -> None:
    """Writes the binary cache of a `*.blutils.json` (include/blu_pipeline.h: blu_db_cache_build); pass the cache file
    wherever a taxonomies file is expected."""
    L = _bind()
    L.blu_db_cache_build.restype = C.c_int
    L.blu_db_cache_build.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    rc = L.blu_db_cache_build(taxonomies_file.encode(), 1 if use_taxid else 0, cache_file.encode())
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_db_cache_build")


def custom_taxon_from_file(path: str) -> dict:
    """CustomTaxon::from_file (domain/dtos/taxon.rs:28-66)."""
    cfg = N.CutoffConfig()
    rc = _bind().blu_custom_taxon_from_file(path.encode(), C.byref(cfg))
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_custom_taxon_from_file")
    return {k: int(cfg.custom[i]) for i, k in enumerate(N.CUSTOM_FIELDS) if cfg.custom_has[i]}


def build_consensus_identities(blast_output: str, taxonomies_file: str, taxon: str = "bacteria",
                               strategy: str = "relaxed", use_taxid: Optional[bool] = None,
                               custom_taxon_values: Optional[dict] = None, headers: Optional[Sequence[str]] = None,
                               out_format: str = "json", device: int = 0, lenient: bool = False, parse: bool = True,
                               config=None, out_path: Optional[str] = None):
    """Returns (results, stats).  With out_path the document is written there by the library (no copy through Python) and
    (None, stats) is returned.  results: the parsed `results` list (json) / list of records (jsonl), sorted by
    query, or the raw text when parse=False.  config: Some(BlastBuilder) of the run-with-consensus path
    (blutils_amd.blast.BlastBuilder): its run id goes on every result and it is written as the document's config."""
    L = _bind()
    p = PipelineParams()
    p.cutoffs.taxon = N.TAXON[taxon]
    p.cutoffs.has_custom = 1 if custom_taxon_values is not None else 0
    if custom_taxon_values is not None:
        for i, k in enumerate(N.CUSTOM_FIELDS):
            if custom_taxon_values.get(k) is not None:
                p.cutoffs.custom[i] = int(custom_taxon_values[k])
                p.cutoffs.custom_has[i] = 1
    p.strategy = N.STRATEGY[strategy]
    p.use_taxid = 1 if use_taxid else 0
    p.device = device
    p.out_format = OUT_FORMAT[out_format]
    p.lenient = 1 if lenient else 0
    hdr_arr, n_hdr = None, 0
    if headers is not None:
        enc = [h.encode() for h in headers]
        hdr_arr = (C.c_char_p * max(1, len(enc)))(*enc)
        n_hdr = len(enc)
    text, n = C.c_void_p(), C.c_size_t()
    st = PipelineStats()
    L.blu_build_consensus_identities_cfg.restype = C.c_int
    L.blu_build_consensus_identities_cfg.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.POINTER(PipelineParams),
                                                     C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                                     C.POINTER(PipelineStats)]
    run_id = str(config.run_id).encode() if config is not None else None
    cfg_text = config.render(out_format).encode() if config is not None else None
    if out_path is not None:
        L.blu_build_consensus_identities_to_file.restype = C.c_int
        L.blu_build_consensus_identities_to_file.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_char_p, C.POINTER(PipelineParams),
                                                             C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(PipelineStats)]
        rc = L.blu_build_consensus_identities_to_file(blast_output.encode(), C.cast(hdr_arr, C.c_void_p) if hdr_arr else None, n_hdr,
                                                      taxonomies_file.encode(), C.byref(p), run_id, cfg_text, out_path.encode(),
                                                      C.byref(st))
        if rc != N.BLU_OK:
            raise N.BluError(rc, "blu_build_consensus_identities_to_file")
        return None, {f: getattr(st, f) for f, _ in PipelineStats._fields_}
    rc = L.blu_build_consensus_identities_cfg(blast_output.encode(), C.cast(hdr_arr, C.c_void_p) if hdr_arr else None, n_hdr,
                                              taxonomies_file.encode(), C.byref(p), run_id, cfg_text, C.byref(text),
                                              C.byref(n), C.byref(st))
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_build_consensus_identities")
    try:
        raw = C.string_at(text, n.value).decode("utf-8")
    finally:
        L.blu_free_text(text)
    stats = {f: getattr(st, f) for f, _ in PipelineStats._fields_}
    if not parse:
        return raw, stats
    if out_format in ("json", "json-compact"):
        return json.loads(raw)["results"], stats
    if out_format == "yaml":
        import yaml
        return yaml.safe_load(raw)["results"], stats
    lines = raw.splitlines()
    assert config is not None or lines[0] == "null"   # the config line comes first (write_blutils_output.rs:169-175)
    return [json.loads(l) for l in lines[1:]], stats
