"""Host-side handle objects over the C ABI (include/blu_consensus.h).

Mirrors the reference seam `build_consensus_identities`
(core/src/use_cases/build_consensus_identities/mod.rs:40-47): a taxonomy + a
cutoff configuration (Taxon, Option<CustomTaxon>) on one side, grouped hit rows
on the other, a strategy, one result per query.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _native as N

RESULT_DTYPE = np.dtype([
    ("status", "u1"), ("flags", "u1"), ("bean_index", "u1"), ("max_allowed_level", "u1"),
    ("reached_rank", "<u2"), ("max_allowed_rank", "<u2"), ("identifier_node", "<u4"), ("ref_row", "<u4"),
    ("level_mask", "<u8"), ("ident_used", "<f8"),
])
assert RESULT_DTYPE.itemsize == 32


def _cutoff_config(taxon: str, custom: Optional[dict]) -> N.CutoffConfig:
    cfg = N.CutoffConfig()
    cfg.taxon = N.TAXON[taxon]
    cfg.has_custom = 1 if custom is not None else 0
    if custom is not None:
        for i, k in enumerate(N.CUSTOM_FIELDS):
            if custom.get(k) is not None:
                cfg.custom[i] = int(custom[k])
                cfg.custom_has[i] = 1
    return cfg


class Taxonomy:
    """Device-resident taxonomy + per-shape cutoff tables (blu_taxonomy_create)."""

    def __init__(self, lin_off, lin_node, lin_rank, rank_names: Sequence[str], taxon: str = "bacteria",
                 custom: Optional[dict] = None, device: int = 0, taxid=None, bad=None):
        L = N.lib()
        self.lin_off = np.ascontiguousarray(lin_off, dtype=np.uint64)
        self.lin_node = np.ascontiguousarray(lin_node, dtype=np.uint32)
        self.lin_rank = np.ascontiguousarray(lin_rank, dtype=np.uint16)
        self.rank_names = list(rank_names)
        self.taxon, self.custom, self.device = taxon, custom, device
        names = [s.encode() for s in self.rank_names]
        arr = (C.c_char_p * max(1, len(names)))(*names)
        desc = N.TaxonomyDesc()
        desc.n_tax = len(self.lin_off) - 1
        desc.lin_off = self.lin_off.ctypes.data
        desc.lin_node = self.lin_node.ctypes.data
        desc.lin_rank = self.lin_rank.ctypes.data
        desc.n_ranks = len(names)
        desc.rank_names = C.cast(arr, C.c_void_p)
        self._taxid = None
        if taxid is not None:
            self._taxid = np.ascontiguousarray(taxid, dtype=np.int64)
            desc.taxid = self._taxid.ctypes.data
        if bad is not None:
            self._bad = np.ascontiguousarray(bad, dtype=np.uint8)
            desc.bad = self._bad.ctypes.data
        cfg = _cutoff_config(taxon, custom)
        h = C.c_void_p()
        rc = L.blu_taxonomy_create(C.byref(desc), C.byref(cfg), device, C.byref(h))
        if rc != N.BLU_OK:
            raise N.BluError(rc, "blu_taxonomy_create")
        self._h = h

    # -- introspection -------------------------------------------------------
    @property
    def handle(self):
        return self._h

    @property
    def n_tax(self) -> int:
        return N.lib().blu_taxonomy_n_tax(self._h)

    @property
    def n_shapes(self) -> int:
        return N.lib().blu_taxonomy_n_shapes(self._h)

    @property
    def max_depth(self) -> int:
        return N.lib().blu_taxonomy_max_depth(self._h)

    @property
    def device_bytes(self) -> int:
        return N.lib().blu_taxonomy_device_bytes(self._h)

    def rank_name(self, code: int, serde: bool = False) -> str:
        p = N.lib().blu_taxonomy_rank_name(self._h, int(code), 1 if serde else 0)
        if p is None:
            raise IndexError(code)
        return p.decode()

    def row_cutoffs(self, row: int):
        cut = np.zeros(64, dtype=np.float64)
        isdef = np.zeros(64, dtype=np.uint8)
        codes = np.zeros(64, dtype=np.uint16)
        n = N.lib().blu_taxonomy_row_cutoffs(self._h, int(row), 64, cut.ctypes.data, isdef.ctypes.data, codes.ctypes.data)
        if n < 0:
            raise IndexError(row)
        return cut[:n], isdef[:n].astype(bool), codes[:n]

    def row_map(self):
        """(desc row -> engine row id, sorted position -> desc row).  Engine row ids (what blu_hits.tax_row holds)
        are opaque: sorted position in lexicographic lineage order | lineage length << 25."""
        if getattr(self, "_row_map", None) is None:
            fwd = np.zeros(max(1, self.n_tax), dtype=np.uint32)
            inv = np.zeros(max(1, self.n_tax), dtype=np.uint32)
            rc = N.lib().blu_taxonomy_row_map(self._h, fwd.ctypes.data, inv.ctypes.data)
            if rc != N.BLU_OK:
                raise N.BluError(rc, "blu_taxonomy_row_map")
            self._row_map = (fwd[: self.n_tax], inv[: self.n_tax])
        return self._row_map

    def engine_rows(self, desc_rows):
        """desc row indices (numpy, or a torch tensor on any device; -1 / 0xFFFFFFFF = unmatched) -> engine row ids."""
        fwd, _ = self.row_map()
        if isinstance(desc_rows, np.ndarray):
            r = desc_rows.view(np.uint32) if desc_rows.dtype == np.int32 else desc_rows.astype(np.uint32)
            ok = r < self.n_tax
            out = np.full(r.shape, N.BLU_UNMATCHED_TAXID, dtype=np.uint32)
            out[ok] = fwd[r[ok]]
            return out
        import torch
        key = str(desc_rows.device)
        cache = self.__dict__.setdefault("_row_map_t", {})
        if key not in cache:
            cache[key] = torch.from_numpy(fwd.view(np.int32).copy()).to(desc_rows.device)   # uint32 bit patterns
        m = cache[key]
        ok = (desc_rows >= 0) & (desc_rows < self.n_tax)
        return torch.where(ok, m[desc_rows.clamp(min=0, max=max(0, self.n_tax - 1)).long()], torch.full_like(desc_rows, -1))

    def lookup(self, taxids) -> np.ndarray:
        t = np.ascontiguousarray(taxids, dtype=np.int64)
        out = np.zeros(len(t), dtype=np.uint32)
        rc = N.lib().blu_taxonomy_lookup(self._h, t.ctypes.data, len(t), out.ctypes.data)
        if rc != N.BLU_OK:
            raise N.BluError(rc, "blu_taxonomy_lookup")
        return out

    def close(self):
        if getattr(self, "_h", None):
            N.lib().blu_taxonomy_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _u32(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.int32 else np.ascontiguousarray(a, dtype=np.uint32)


def pack_records(tax: "Taxonomy", tax_row, pident_milli, align_len, acc_rank, pident=None, wide: bool = False) -> np.ndarray:
    """blu_hits_pack / blu_hits_pack64 on host arrays: the [n, 4] (or, wide, [n, 6]) uint32 side records of the packed
    layouts (include/blu_consensus.h) — {tax_row, pident_milli | shape hint << 17, align_len, acc_rank} (+ the f64
    perc_identity for the wide form).  tax_row: ENGINE row ids."""
    tx, aln, ac = _u32(tax_row), np.ascontiguousarray(align_len, dtype=np.int32), _u32(acc_rank)
    pm = _u32(pident_milli) if pident_milli is not None else None
    pid = np.ascontiguousarray(pident, dtype=np.float64) if pident is not None else None
    n = len(tx)
    out = np.zeros((n, 6 if wide else 4), dtype=np.uint32)
    cols = N.Hits(None, tx.ctypes.data, pid.ctypes.data if pid is not None else None, aln.ctypes.data, ac.ctypes.data, None, n, 0, 0, 0,
                  pm.ctypes.data if pm is not None else None, None, None)
    fn = N.lib().blu_hits_pack64 if wide else N.lib().blu_hits_pack
    rc = fn(tax.handle, C.byref(cols), out.ctypes.data, None)
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_hits_pack")
    return out


def pack_hits_device(tax: "Taxonomy", hits: dict, wide: bool = False):
    """blu_hits_pack / blu_hits_pack64 on torch CUDA columns (keys tax_row, align_len, acc_rank and pident_milli or pident):
    the int32 tensor of 4 (wide: 6) words per hit that goes into `packed` / `packed64`."""
    import torch
    n = hits["tax_row"].numel()
    out = torch.empty((6 if wide else 4) * n, dtype=torch.int32, device=hits["tax_row"].device)
    pm, pid = hits.get("pident_milli"), hits.get("pident")
    if pm is not None:
        pid = None
    for k in ("tax_row", "align_len", "acc_rank"):
        assert hits[k].is_cuda and hits[k].is_contiguous() and hits[k].dtype == torch.int32, k
    cols = N.Hits(None, hits["tax_row"].data_ptr(), pid.data_ptr() if pid is not None else None, hits["align_len"].data_ptr(),
                  hits["acc_rank"].data_ptr(), None, n, 0, 1, 0, pm.data_ptr() if pm is not None else None, None, None)
    fn = N.lib().blu_hits_pack64 if wide else N.lib().blu_hits_pack
    rc = fn(tax.handle, C.byref(cols), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_hits_pack")
    return out


def run_consensus_host(tax: Taxonomy, seg_off, bitscore, tax_row, pident, align_len, acc_rank,
                       strategy: str = "relaxed", pident_milli=None, packed=False) -> np.ndarray:
    """Host buffers in, host records out; the library stages them over PCIe.  tax_row: ENGINE row ids.
    pident_milli (uint32, perc_identity * 1000) replaces the f64 `pident` column when given (pass pident=None);
    packed=True hands the four non-bit-score columns over as 16-byte records built by blu_hits_pack (pident_milli, or an
    f64 column of exact milli-percent values); packed="wide": the 24-byte records of blu_hits_pack64 (any f64)."""
    if packed:
        seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
        bs = np.ascontiguousarray(bitscore, dtype=np.int32)
        wide = packed == "wide"
        rec = pack_records(tax, tax_row, pident_milli, align_len, acc_rank, pident=pident if pident_milli is None else None, wide=wide)
        nq, nh = len(seg) - 1, int(seg[-1])
        assert rec.shape == (nh, 6 if wide else 4) and rec.ctypes.data % 16 == 0
        hits = N.Hits(bs.ctypes.data, None, None, None, None, seg.ctypes.data, nh, nq, 0, 0, None, None if wide else rec.ctypes.data,
                      rec.ctypes.data if wide else None)
        params = N.RunParams(N.STRATEGY[strategy], 0, None)
        out = np.zeros(nq, dtype=RESULT_DTYPE)
        rc = N.lib().blu_consensus_run(tax.handle, C.byref(hits), C.byref(params), out.ctypes.data)
        if rc != N.BLU_OK:
            raise N.BluError(rc, "blu_consensus_run")
        return out
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    bs = np.ascontiguousarray(bitscore, dtype=np.int32)
    tx = np.ascontiguousarray(tax_row)
    tx = tx.view(np.uint32) if tx.dtype == np.int32 else np.ascontiguousarray(tx, dtype=np.uint32)
    pid = np.ascontiguousarray(pident, dtype=np.float64) if pident_milli is None else None
    pm = None
    if pident_milli is not None:
        pm = np.ascontiguousarray(pident_milli)
        pm = pm.view(np.uint32) if pm.dtype == np.int32 else np.ascontiguousarray(pm, dtype=np.uint32)
    aln = np.ascontiguousarray(align_len, dtype=np.int32)
    ac = np.ascontiguousarray(acc_rank)
    ac = ac.view(np.uint32) if ac.dtype == np.int32 else np.ascontiguousarray(ac, dtype=np.uint32)
    nq = len(seg) - 1
    nh = int(seg[-1])
    assert len(bs) == nh and len(tx) == nh and len(pid if pm is None else pm) == nh and len(aln) == nh and len(ac) == nh
    hits = N.Hits(bs.ctypes.data, tx.ctypes.data, pid.ctypes.data if pm is None else None, aln.ctypes.data, ac.ctypes.data,
                  seg.ctypes.data, nh, nq, 0, 0, pm.ctypes.data if pm is not None else None, None, None)
    params = N.RunParams(N.STRATEGY[strategy], 0, None)
    out = np.zeros(nq, dtype=RESULT_DTYPE)
    rc = N.lib().blu_consensus_run(tax.handle, C.byref(hits), C.byref(params), out.ctypes.data)
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_consensus_run")
    return out


def shard_ranges(seg_off, n_shards: int) -> np.ndarray:
    """blu_shard_ranges: query bounds of n_shards contiguous ranges balanced by hit count."""
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    bounds = np.zeros(n_shards + 1, dtype=np.uint64)
    rc = N.lib().blu_shard_ranges(seg.ctypes.data_as(C.c_void_p), C.c_uint64(len(seg) - 1), C.c_uint32(n_shards),
                                  bounds.ctypes.data_as(C.c_void_p))
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_shard_ranges")
    return bounds


def run_consensus_multi(taxes: Sequence[Taxonomy], seg_off, bitscore, tax_row, pident, align_len, acc_rank,
                        strategy: str = "relaxed", pident_milli=None) -> np.ndarray:
    """blu_consensus_run_multi: one host table over several handles of the same taxonomy (one per GPU); `tax_row` holds
    engine row ids (identical for every handle of one taxonomy)."""
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    bs = np.ascontiguousarray(bitscore, dtype=np.int32)
    tx = np.ascontiguousarray(tax_row)
    tx = tx.view(np.uint32) if tx.dtype == np.int32 else np.ascontiguousarray(tx, dtype=np.uint32)
    pid = np.ascontiguousarray(pident, dtype=np.float64) if pident_milli is None else None
    pm = np.ascontiguousarray(pident_milli, dtype=np.uint32) if pident_milli is not None else None
    aln = np.ascontiguousarray(align_len, dtype=np.int32)
    ac = np.ascontiguousarray(acc_rank)
    ac = ac.view(np.uint32) if ac.dtype == np.int32 else np.ascontiguousarray(ac, dtype=np.uint32)
    nq, nh = len(seg) - 1, int(seg[-1])
    hits = N.Hits(bs.ctypes.data, tx.ctypes.data, pid.ctypes.data if pm is None else None, aln.ctypes.data, ac.ctypes.data,
                  seg.ctypes.data, nh, nq, 0, 0, pm.ctypes.data if pm is not None else None, None, None)
    params = N.RunParams(N.STRATEGY[strategy], 0, None)
    out = np.zeros(nq, dtype=RESULT_DTYPE)
    handles = (C.c_void_p * len(taxes))(*[t.handle for t in taxes])
    L = N.lib()
    L.blu_consensus_run_multi.restype = C.c_int
    L.blu_consensus_run_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.blu_consensus_run_multi(handles, len(taxes), C.byref(hits), C.byref(params), out.ctypes.data)
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_consensus_run_multi")
    return out


def run_consensus_device(tax: Taxonomy, hits: dict, out, strategy: str = "relaxed", stream: Optional[int] = None):
    """torch CUDA tensors in (`hits` keys: seg_off bitscore tax_row align_len acc_rank and either pident (float64) or
    pident_milli (int32 bit pattern of uint32)), records into the uint8 CUDA tensor `out` of 32 * n_queries bytes.
    Asynchronous on `stream` (default: torch's current stream)."""
    import torch

    nq = hits["seg_off"].numel() - 1
    nh = hits["bitscore"].numel()
    milli = hits.get("pident_milli") is not None
    packed = hits.get("packed") is not None
    wide = hits.get("packed64") is not None
    if packed or wide:     # side records next to the bit-score column (blu_hits_pack / blu_hits_pack64)
        want = (("seg_off", torch.int64), ("bitscore", torch.int32), ("packed64" if wide else "packed", torch.int32))
    else:
        want = (("seg_off", torch.int64), ("bitscore", torch.int32), ("tax_row", torch.int32),
                ("pident_milli", torch.int32) if milli else ("pident", torch.float64), ("align_len", torch.int32),
                ("acc_rank", torch.int32))
    for k, dt in want:
        t = hits[k]
        assert t.is_cuda and t.is_contiguous() and t.dtype == dt, (k, t.dtype, t.device)
    assert out.is_cuda and out.is_contiguous() and out.numel() * out.element_size() >= 32 * nq
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    if wide:
        assert hits["packed64"].numel() == 6 * nh and hits["packed64"].data_ptr() % 8 == 0
        h = N.Hits(hits["bitscore"].data_ptr(), None, None, None, None, hits["seg_off"].data_ptr(), nh, nq, 1, 0, None, None,
                   hits["packed64"].data_ptr())
    elif packed:
        assert hits["packed"].numel() == 4 * nh and hits["packed"].data_ptr() % 16 == 0
        h = N.Hits(hits["bitscore"].data_ptr(), None, None, None, None, hits["seg_off"].data_ptr(), nh, nq, 1, 0, None,
                   hits["packed"].data_ptr(), None)
    else:
        h = N.Hits(hits["bitscore"].data_ptr(), hits["tax_row"].data_ptr(), None if milli else hits["pident"].data_ptr(),
                   hits["align_len"].data_ptr(), hits["acc_rank"].data_ptr(), hits["seg_off"].data_ptr(), nh, nq, 1, 0,
                   hits["pident_milli"].data_ptr() if milli else None, None, None)
    params = N.RunParams(N.STRATEGY[strategy], 0, stream)
    rc = N.lib().blu_consensus_run(tax.handle, C.byref(h), C.byref(params), out.data_ptr())
    if rc != N.BLU_OK:
        raise N.BluError(rc, "blu_consensus_run")


def records_from_tensor(out) -> np.ndarray:
    """uint8 CUDA/CPU tensor -> numpy structured array of blu_result."""
    return out.detach().cpu().numpy().view(np.uint8).reshape(-1)[: (out.numel() * out.element_size()) // 32 * 32].view(RESULT_DTYPE)


def last_launch():
    name = C.create_string_buffer(128)
    grid, block = C.c_uint32(), C.c_uint32()
    N.lib().blu_consensus_last_launch(name, 128, C.byref(grid), C.byref(block))
    return name.value.decode(), grid.value, block.value
