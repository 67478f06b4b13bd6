"""ctypes binding of libblu_consensus.so (include/blu_consensus.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLU_CONSENSUS_LIB: load an experimental build of the same ABI (kernel A/B experiments under scripts/)
LIB_PATH = os.environ.get("BLU_CONSENSUS_LIB") or os.path.join(_HERE, "lib", "libblu_consensus.so")

# every symbol include/blu_consensus.h declares
EXPORTS = (
    "blu_abi_version", "blu_last_error", "blu_consensus_run_multi", "blu_shard_ranges", "blu_taxonomy_create", "blu_taxonomy_destroy", "blu_taxonomy_n_tax",
    "blu_taxonomy_n_shapes", "blu_taxonomy_n_rank_codes", "blu_taxonomy_max_depth", "blu_taxonomy_device_bytes",
    "blu_taxonomy_rank_name", "blu_taxonomy_row_cutoffs", "blu_taxonomy_lookup", "blu_taxonomy_row_map", "blu_consensus_run",
    "blu_consensus_last_launch", "blu_hits_pack", "blu_hits_pack64", "blu_taxonomy_shared_levels", "blu_taxonomy_trim",
)
# include/blu_pipeline.h
PIPELINE_EXPORTS = ("blu_build_consensus_identities", "blu_free_text", "blu_custom_taxon_from_file", "blu_ingest_only",
                    "blu_db_cache_build", "blu_build_consensus_identities_cfg", "blu_ingest_only_on", "blu_last_ingest_path",
                    "blu_build_consensus_identities_to_file", "blu_ingest_columns_on", "blu_ingest_columns_free")

BLU_UNMATCHED_TAXID = 0xFFFFFFFF
BLU_NONE_U8, BLU_NONE_U16, BLU_MAR_NEVER_EQUAL = 0xFF, 0xFFFF, 0xFFFE
BLU_OK, BLU_ERR_INVALID_ARG, BLU_ERR_NO_DEVICE, BLU_ERR_HIP, BLU_ERR_DEPTH, BLU_ERR_CUSTOM_MISSING = 0, 1, 2, 3, 4, 5
ST_MULTI, ST_SINGLE, ST_NO_HITS = 0, 1, 2
ST_ERR_UNMATCHED, ST_ERR_BAD_LINEAGE, ST_ERR_ROOT, ST_ERR_SINGLE_BELOW, ST_ERR_BAD_PIDENT = 16, 17, 18, 19, 20
FLAG_MUTATED, FLAG_AGREE = 1, 2
TAXON = {"fungi": 0, "bacteria": 1, "eukaryotes": 2, "custom": 3}
STRATEGY = {"cautious": 0, "relaxed": 1}
CUSTOM_FIELDS = ("domain", "kingdom", "phylum", "class", "order", "family", "genus", "species")


class CutoffConfig(C.Structure):
    _fields_ = [("taxon", C.c_int32), ("has_custom", C.c_int32), ("custom", C.c_int16 * 8),
                ("custom_has", C.c_uint8 * 8)]


class TaxonomyDesc(C.Structure):
    _fields_ = [("n_tax", C.c_uint64), ("taxid", C.c_void_p), ("lin_off", C.c_void_p), ("lin_node", C.c_void_p),
                ("lin_rank", C.c_void_p), ("n_ranks", C.c_uint32), ("rank_names", C.c_void_p), ("bad", C.c_void_p)]


class Hits(C.Structure):
    _fields_ = [("bitscore", C.c_void_p), ("tax_row", C.c_void_p), ("pident", C.c_void_p), ("align_len", C.c_void_p),
                ("acc_rank", C.c_void_p), ("seg_off", C.c_void_p), ("n_hits", C.c_uint64), ("n_queries", C.c_uint64),
                ("on_device", C.c_int32), ("reserved", C.c_int32), ("pident_milli", C.c_void_p), ("packed", C.c_void_p),
                ("packed64", C.c_void_p)]


class RunParams(C.Structure):
    _fields_ = [("strategy", C.c_int32), ("flags", C.c_int32), ("stream", C.c_void_p)]


class NativeLibraryMissing(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Loads the HIP library; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C blutils_amd/csrc).  blutils_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.blu_abi_version.restype = C.c_uint32
    L.blu_last_error.restype = C.c_size_t
    L.blu_last_error.argtypes = [C.c_char_p, C.c_size_t]
    L.blu_taxonomy_create.restype = C.c_int
    L.blu_taxonomy_create.argtypes = [C.POINTER(TaxonomyDesc), C.POINTER(CutoffConfig), C.c_int, C.POINTER(C.c_void_p)]
    L.blu_taxonomy_destroy.argtypes = [C.c_void_p]
    for name, rt in (("blu_taxonomy_n_tax", C.c_uint64), ("blu_taxonomy_n_shapes", C.c_uint32),
                     ("blu_taxonomy_n_rank_codes", C.c_uint32), ("blu_taxonomy_max_depth", C.c_uint32),
                     ("blu_taxonomy_device_bytes", C.c_uint64)):
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = [C.c_void_p]
    L.blu_taxonomy_rank_name.restype = C.c_char_p
    L.blu_taxonomy_rank_name.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.blu_taxonomy_row_cutoffs.restype = C.c_int32
    L.blu_taxonomy_row_cutoffs.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.blu_taxonomy_lookup.restype = C.c_int
    L.blu_taxonomy_lookup.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.blu_taxonomy_row_map.restype = C.c_int
    L.blu_taxonomy_row_map.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.blu_consensus_run.restype = C.c_int
    L.blu_consensus_run.argtypes = [C.c_void_p, C.POINTER(Hits), C.POINTER(RunParams), C.c_void_p]
    for name in ("blu_hits_pack", "blu_hits_pack64"):
        if not hasattr(L, name):                 # (an A/B library of an older ABI: BLU_CONSENSUS_LIB, scripts/ab.sh)
            continue
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [C.c_void_p, C.POINTER(Hits), C.c_void_p, C.c_void_p]
    if hasattr(L, "blu_taxonomy_shared_levels"):
        L.blu_taxonomy_shared_levels.restype = C.c_int
        L.blu_taxonomy_shared_levels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        L.blu_taxonomy_trim.restype = C.c_int
        L.blu_taxonomy_trim.argtypes = [C.c_void_p]
    L.blu_consensus_last_launch.restype = C.c_int
    L.blu_consensus_last_launch.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    _lib = L
    return L


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().blu_last_error(buf, 1024)
    return buf.value.decode("utf-8", "replace")


class BluError(RuntimeError):
    def __init__(self, code: int, where: str):
        self.code = code
        super().__init__(f"{where} failed with blu_error {code}: {last_error()}")
