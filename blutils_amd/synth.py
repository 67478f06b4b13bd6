"""Synthetic workloads of BASELINE.json's configs.

* `make_taxonomy`  — random rooted taxonomy with NCBI-like rank structure
  (backbone d,k,p,c,o,f,g,s; optional cellular-root, clades, species-group /
  species-subgroup, strain/subspecies/no-rank tails, lineages ending early at
  g or f), returned as the CSR arrays blu_taxonomy_create consumes.  numpy, CPU.
* `make_hits`      — outfmt-6-like hit table straight into the SoA columns, from
  a counter-based splitmix64 stream written with torch integer ops, so the same
  seed gives the same table on CPU and on the GPU.

Seeds per config: C1 0xB10751, C2 0xB10752, C3/C4 0xB10753, C5 0xB10755.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

GENERATOR_VERSION = 1
SEEDS = {"C1": 0xB10751, "C2": 0xB10752, "C3": 0xB10753, "C4": 0xB10753, "C5": 0xB10755}

RANK_NAMES = ["u", "d", "k", "p", "c", "o", "f", "g", "s", "clade", "cellular-root", "species-group",
              "species-subgroup", "strain", "subspecies", "no-rank"]
_R = {n: i for i, n in enumerate(RANK_NAMES)}
BACKBONE = ["d", "k", "p", "c", "o", "f", "g", "s"]
UNMATCHED_I32 = -1  # 0xFFFFFFFF


# ----------------------------------------------------------------------------
# splitmix64 on numpy uint64 (taxonomy side)
# ----------------------------------------------------------------------------
def _mix_np(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _h_np(seed: int, salt: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        k = idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64((salt * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)
    return _mix_np(_mix_np(k) ^ np.uint64(seed & 0xFFFFFFFFFFFFFFFF))


@dataclass
class SynthTaxonomy:
    rank_names: List[str]
    lin_off: np.ndarray     # uint64 [n+1]
    lin_node: np.ndarray    # uint32
    lin_rank: np.ndarray    # uint16 (index into rank_names)
    taxid: np.ndarray       # int64 [n]
    level_lo: np.ndarray    # int32 [9, n]  subject range sharing the level-r ancestor (row 0 = whole table)
    level_hi: np.ndarray    # int32 [9, n]
    n: int = 0
    seed: int = 0
    deep: bool = False

    def lineage_strings(self, text: bool = True) -> List[str]:
        """`rank__identifier;...` strings (blutils DB grammar, build_taxonomy_database.rs:406-465)."""
        names = self.rank_names
        off = self.lin_off
        node = self.lin_node
        rank = self.lin_rank
        pre = "n" if text else ""
        out = []
        for t in range(self.n):
            a, b = int(off[t]), int(off[t + 1])
            out.append(";".join(f"{names[rank[i]]}__{pre}{node[i]}" for i in range(a, b)))
        return out


def make_taxonomy(n_taxa: int, seed: int, deep: bool = False) -> SynthTaxonomy:
    """Random taxonomy with `n_taxa` leaf taxids in DFS order.

    deep=True is the C5 flavour: cellular-root everywhere, clades with p=0.6 and
    up to 3 stacked (lineage depth 25-40).
    """
    n = int(n_taxa)
    rng = np.random.default_rng(seed)
    # group counts per backbone level: species = every taxon, then /12 /8 /5 /4 /4 /6 /4 upwards
    fans = [12, 8, 5, 4, 4, 6, 4]
    counts = [n]
    for f in fans:
        counts.append(max(1, counts[-1] // f))
    counts = counts[::-1]                      # d .. s
    counts[0] = min(max(counts[0], 3), n)      # a few domains so that level-0 disagreement exists
    for r in range(1, 8):
        counts[r] = min(n, max(counts[r], counts[r - 1]))
    bounds = []
    cuts = np.zeros(0, dtype=np.int64)
    for r in range(8):
        need = counts[r] - 1 - len(cuts)
        if r == 7:
            cuts = np.arange(1, n, dtype=np.int64)
        elif need > 0:
            pool = np.setdiff1d(np.arange(1, n, dtype=np.int64), cuts, assume_unique=False)
            extra = rng.choice(pool, size=min(need, len(pool)), replace=False)
            cuts = np.union1d(cuts, extra)
        bounds.append(np.concatenate([[0], cuts, [n]]).astype(np.int64))
    t = np.arange(n, dtype=np.int64)
    gid = [np.searchsorted(b, t, side="right") - 1 for b in bounds]
    level_lo = np.zeros((9, n), dtype=np.int32)
    level_hi = np.full((9, n), n, dtype=np.int32)
    for r in range(8):
        level_lo[r + 1] = bounds[r][gid[r]]
        level_hi[r + 1] = bounds[r][gid[r] + 1]

    # columns of the lineage matrix; -1 = absent
    p_a = 0.6 if deep else 0.15
    p_b = 0.3 if deep else 0.10
    max_stack = 3 if deep else 1
    cols_id: List[np.ndarray] = []
    cols_rank: List[np.ndarray] = []
    bb_col = []  # column index of each backbone level

    # every column owns a disjoint id block (base .. base + number of groups), so ids are unique
    # (rank, identifier) nodes without a sort/unique pass; -1 = level absent for that taxon
    next_base = [0]

    def key(kind: int, r: int, g: np.ndarray) -> np.ndarray:
        base = next_base[0]
        next_base[0] += int(g.max()) + 1 if len(g) else 0
        return (g + base).astype(np.int64)

    absent = np.full(n, -1, dtype=np.int64)
    for r in range(8):
        g = gid[r]
        if r == 0:
            # cellular-root above a domain (per domain; always for the deep flavour)
            u = _h_np(seed, 100, g) % np.uint64(1000)
            on = (u < (1000 if deep else 500))
            cols_id.append(np.where(on, key(1, 0, g), absent))
            cols_rank.append(np.full(n, _R["cellular-root"], dtype=np.int16))
        else:
            # B: one clade shared by all children of the parent node
            gp = gid[r - 1]
            ub = _h_np(seed, 200 + r, gp) % np.uint64(1000)
            onb = ub < int(p_b * 1000)
            cols_id.append(np.where(onb, key(2, r, gp), absent))
            cols_rank.append(np.full(n, _R["clade"], dtype=np.int16))
            # A: private intermediate nodes above this node
            ua = _h_np(seed, 300 + r, g)
            if r == 7:
                sel = ua % np.uint64(1000)
                grp = sel < 100          # species-group
                sub = sel < 50           # species-group + species-subgroup
                cols_id.append(np.where(grp, key(3, r, g), absent))
                cols_rank.append(np.full(n, _R["species-group"], dtype=np.int16))
                cols_id.append(np.where(sub, key(4, r, g), absent))
                cols_rank.append(np.full(n, _R["species-subgroup"], dtype=np.int16))
            else:
                for s in range(max_stack):
                    us = (ua >> np.uint64(10 * s)) % np.uint64(1000)
                    on = us < int(p_a * 1000)
                    if s > 0:
                        on &= prev_on
                    prev_on = on
                    cols_id.append(np.where(on, key(5 + s, r, g), absent))
                    cols_rank.append(np.full(n, _R["clade"], dtype=np.int16))
        bb_col.append(len(cols_id))
        cols_id.append(key(0, r, g))
        cols_rank.append(np.full(n, _R[BACKBONE[r]], dtype=np.int16))
    # below-species tail: private strain / subspecies / no-rank
    ut = _h_np(seed, 400, t) % np.uint64(1000)
    tail = ut < 100
    tail_rank = np.where(ut < 40, _R["strain"], np.where(ut < 80, _R["subspecies"], _R["no-rank"])).astype(np.int16)
    cols_id.append(np.where(tail, key(9, 0, t), absent))
    cols_rank.append(tail_rank)
    assert next_base[0] < (1 << 32)

    ncol = len(cols_id)
    # 5 % of taxids end early at g or f (the taxid IS that node)
    ue = _h_np(seed, 500, t) % np.uint64(1000)
    last_col = np.full(n, ncol - 1, dtype=np.int64)
    last_col = np.where(ue < 30, bb_col[6], last_col)
    last_col = np.where((ue >= 30) & (ue < 50), bb_col[5], last_col)
    keep = np.empty((n, ncol), dtype=bool)
    ids = np.empty((n, ncol), dtype=np.uint32)
    rks = np.empty((n, ncol), dtype=np.uint16)
    for c in range(ncol):
        keep[:, c] = (cols_id[c] >= 0) & (c <= last_col)
        ids[:, c] = cols_id[c].astype(np.uint32, copy=False)
        rks[:, c] = cols_rank[c]
        cols_id[c] = None
    lens = keep.sum(axis=1, dtype=np.int64)
    assert lens.max() <= 64, "lineage deeper than 64 levels"
    dense = ids[keep]
    flat_rk = rks[keep]
    lin_off = np.zeros(n + 1, dtype=np.uint64)
    lin_off[1:] = np.cumsum(lens)
    taxid = (1000 + 7 * t).astype(np.int64)
    return SynthTaxonomy(RANK_NAMES, lin_off, dense, flat_rk, taxid, level_lo, level_hi, n, seed, deep)


# ----------------------------------------------------------------------------
# splitmix64 with torch int64 ops (hit side; identical on CPU and GPU)
# ----------------------------------------------------------------------------
def _i64(v: int) -> int:
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(x, k: int):
    return (x >> k) & ((1 << (64 - k)) - 1)


def _mix_t(x):
    x = (x ^ _lsr(x, 30)) * _i64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _i64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _h_t(seed: int, salt: int, idx):
    k = idx * _i64(0x9E3779B97F4A7C15) + _i64(salt * 0xD1B54A32D192ED03)
    return _mix_t(_mix_t(k) ^ _i64(seed)) & 0x7FFFFFFFFFFFFFFF   # non-negative


_ZIPF_CACHE: Dict[tuple, np.ndarray] = {}


def _zipf_thresholds(s: float, lo: int, hi: int) -> np.ndarray:
    k = (s, lo, hi)
    if k not in _ZIPF_CACHE:
        h = np.arange(lo, hi + 1, dtype=np.float64)
        w = h ** (-s)
        cdf = np.cumsum(w) / w.sum()
        _ZIPF_CACHE[k] = np.minimum((cdf * float(1 << 62)).astype(np.int64), (1 << 62) - 1)
    return _ZIPF_CACHE[k]


@dataclass
class SynthHits:
    seg_off: "object"     # int64 [Q+1] (bit pattern of uint64)
    bitscore: "object"    # int32
    tax_row: "object"     # int32 (bit pattern of uint32, -1 = BLU_UNMATCHED_TAXID)
    pident: "object"      # float64 (None when only the milli-percent column was generated)
    align_len: "object"   # int32
    acc_rank: "object"    # int32 (bit pattern of uint32)
    n_queries: int = 0
    n_hits: int = 0
    pident_milli: "object" = None   # int32: perc_identity * 1000 (the generator draws 3-decimal values)

    def as_dict(self, layout: str = "f64", tax=None):
        """layout "f64": the canonical 24 B/hit columns; "milli": pident as milli-percent uint32, 20 B/hit; "packed" /
        "packed64": the bit-score column + the side records blu_hits_pack / blu_hits_pack64 build from the columns (tax =
        the engine.Taxonomy whose row ids self.tax_row holds; without it the 16-byte records are put together here, with
        no shape hint — a form the engine also takes)."""
        d = {"seg_off": self.seg_off, "bitscore": self.bitscore, "tax_row": self.tax_row,
             "align_len": self.align_len, "acc_rank": self.acc_rank}
        if layout in ("packed", "packed64"):
            import torch
            from . import _native
            if layout == "packed" and not hasattr(_native.lib(), "blu_hits_pack"):
                tax = None                               # (an A/B library of an older ABI: records without hints)
            if tax is not None and self.tax_row.is_cuda:
                from . import engine
                cols = dict(d)
                if layout == "packed64" and self.pident is not None:
                    cols["pident"] = self.pident
                else:
                    cols["pident_milli"] = self.pident_milli
                return {"seg_off": self.seg_off, "bitscore": self.bitscore, layout: engine.pack_hits_device(tax, cols, wide=layout == "packed64")}
            assert layout == "packed", "packed64 records are built by blu_hits_pack64: pass tax="
            rec = torch.stack([self.tax_row, self.pident_milli, self.align_len, self.acc_rank], dim=1).contiguous().reshape(-1)
            return {"seg_off": self.seg_off, "bitscore": self.bitscore, "packed": rec}
        if layout == "milli":
            d["pident_milli"] = self.pident_milli
        else:
            d["pident"] = self.pident
        return d

    def numpy(self):
        return {k: v.detach().cpu().numpy() for k, v in self.as_dict().items()}

    def algorithmic_bytes(self, layout: str = "f64") -> int:
        """SURVEY §8d / BASELINE.md §3: (24 | 20)·H + 8·(Q+1) + 32·Q — the bytes of the layout actually read."""
        return (24 if layout == "f64" else 20) * self.n_hits + 8 * (self.n_queries + 1) + 32 * self.n_queries


def hit_counts(n_queries: int, seed: int, hits_per_query: Optional[int], zipf: Optional[tuple], device, q_offset: int = 0):
    """Hits per query (int64 tensor) of queries q_offset .. q_offset + n_queries of the table make_hits generates."""
    import torch
    dev = torch.device(device)
    Q = int(n_queries)
    if zipf is None:
        return torch.full((Q,), int(hits_per_query), dtype=torch.int64, device=dev)
    qi = torch.arange(Q, dtype=torch.int64, device=dev) + int(q_offset)
    s, lo, hi = zipf
    thr = torch.from_numpy(_zipf_thresholds(float(s), int(lo), int(hi))).to(dev)
    u = _h_t(seed, 1, qi) & ((1 << 62) - 1)
    return torch.searchsorted(thr, u, right=True).clamp_(max=int(hi - lo)) + int(lo)


def make_hits(tax: SynthTaxonomy, n_queries: int, seed: int, hits_per_query: Optional[int] = 50,
              zipf: Optional[tuple] = None, device: str = "cpu", p_unmatched: float = 0.0005,
              chunk_queries: int = 1 << 20, q_offset: int = 0, tables=None, columns: str = "both",
              top_group: str = "geo") -> SynthHits:
    """Hit table in SoA form.  hits_per_query fixed, or zipf=(s, lo, hi) for the skewed config.

    q_offset shifts the query counter (rank r of a multi-GPU run generates its own slice of one
    global table).  `tables` caches the device copies of the taxonomy range tables.
    top_group: size of the top bit-score group — "geo" = 1 + Geometric(0.35) (mean 2.86); "zymo" = the
    histogram of the reference's one real output (test/mock/output/zymo-mock/blutils.consensus.json: sum of
    `occurrences` over the 2283 results, mean 5.74, 1.3 % single hits); "all" = every hit of the query ties on the
    top score (identical database sequences: the table must be read in full).
    """
    import torch

    dev = torch.device(device)
    Q = int(n_queries)
    n_tax = tax.n
    qi = torch.arange(Q, dtype=torch.int64, device=dev) + int(q_offset)
    nq = hit_counts(Q, seed, hits_per_query, zipf, dev, q_offset)
    seg = torch.zeros(Q + 1, dtype=torch.int64, device=dev)
    torch.cumsum(nq, 0, out=seg[1:])
    H = int(seg[-1].item())
    assert H < (1 << 32) - 1, "n_hits must stay below 2^32 - 1 per call"
    # columns: "both" (f64 and milli-percent pident), "f64" or "milli" (saves 4 / 8 bytes per hit of device memory)
    out = SynthHits(seg, torch.empty(H, dtype=torch.int32, device=dev), torch.empty(H, dtype=torch.int32, device=dev),
                    torch.empty(H, dtype=torch.float64, device=dev) if columns != "milli" else None,
                    torch.empty(H, dtype=torch.int32, device=dev), torch.empty(H, dtype=torch.int32, device=dev), Q, H,
                    torch.empty(H, dtype=torch.int32, device=dev) if columns != "f64" else None)
    if tables is None:
        tables = {}
    if "lo" not in tables or tables["lo"].device != dev:
        tables["lo"] = torch.from_numpy(tax.level_lo).to(dev)
        tables["hi"] = torch.from_numpy(tax.level_hi).to(dev)
        # k / 1000 as the correctly rounded double (= what parsing the 3-decimal text gives); the GPU's
        # own f64 division is not correctly rounded, so the quotients come from a CPU-built table
        tables["pid"] = torch.from_numpy(np.arange(0, 100001, dtype=np.float64) / 1000.0).to(dev)
    lo_t, hi_t, pid_t = tables["lo"], tables["hi"], tables["pid"]
    # LCA-level categorical (per mille): root 5, d 10, k 15, p 20, c 50, o 100, f 150, g 300, s 350
    lvl_thr = torch.tensor([5, 15, 30, 50, 100, 200, 350, 650], dtype=torch.int64, device=dev)
    # geometric(0.35) top-group size: P(size > j) = 0.65^j, as exact integer thresholds on 2^62
    geo = torch.tensor([int((0.65 ** j) * (1 << 62)) for j in range(1, 64)], dtype=torch.int64, device=dev).flip(0)
    forced = torch.tensor([97000, 99000, 66667, 45500], dtype=torch.int64, device=dev)   # 45.5 is below every cutoff
    if top_group == "zymo":
        zsize = torch.tensor(sorted(ZYMO_TOP_GROUPS), dtype=torch.int64, device=dev)
        zcnt = np.array([ZYMO_TOP_GROUPS[k] for k in sorted(ZYMO_TOP_GROUPS)], dtype=np.float64)
        zthr = torch.from_numpy(np.minimum((np.cumsum(zcnt) / zcnt.sum() * float(1 << 62)).astype(np.int64), (1 << 62) - 1)).to(dev)
    elif top_group not in ("geo", "all"):
        raise ValueError(f"unknown top_group {top_group!r}")

    for q0 in range(0, Q, chunk_queries):
        q1 = min(Q, q0 + chunk_queries)
        qq = qi[q0:q1]
        n_c = nq[q0:q1]
        r0, r1 = int(seg[q0].item()), int(seg[q1].item())
        if r1 == r0:
            continue
        if zipf is None:
            qrow = torch.arange(q1 - q0, dtype=torch.int64, device=dev).repeat_interleave(int(hits_per_query))
        else:
            qrow = torch.repeat_interleave(torch.arange(q1 - q0, dtype=torch.int64, device=dev), n_c)
        grow = torch.arange(r0, r1, dtype=torch.int64, device=dev)           # row index inside this table
        gkey = grow + (int(q_offset) << 20)                                   # decorrelate slices of a global table
        j = grow - seg[q0:q1][qrow]                                           # rank inside the query
        # per-query draws
        anchor = _h_t(seed, 2, qq) % n_tax
        lvl = torch.searchsorted(lvl_thr, _h_t(seed, 3, qq) % 1000, right=True)   # 0 = whole table, 1..8 = d..s
        lo = lo_t[lvl, anchor].to(torch.int64)
        hi = hi_t[lvl, anchor].to(torch.int64)
        if top_group == "geo":
            gsz = 63 - torch.searchsorted(geo, _h_t(seed, 4, qq) & ((1 << 62) - 1), right=True) + 1
        elif top_group == "zymo":
            gsz = zsize[torch.searchsorted(zthr, _h_t(seed, 4, qq) & ((1 << 62) - 1), right=True).clamp_(max=len(zsize) - 1)]
        else:
            gsz = n_c.clone()
        gsz = torch.minimum(gsz.clamp_(min=1), n_c)
        B = 200 + _h_t(seed, 5, qq) % 1801
        off = _h_t(seed, 6, qq) % n_c
        share = (_h_t(seed, 7, qq) % 1000) < 300
        share_aln = share & ((_h_t(seed, 8, qq) % 2) == 0)
        q_pid = 80000 + _h_t(seed, 9, qq) % 20001
        q_aln = 380 + _h_t(seed, 10, qq) % 101
        # per-row draws
        span = (hi - lo)[qrow]
        subj = lo[qrow] + _h_t(seed, 11, gkey) % span
        is_top = ((j + off[qrow]) % n_c[qrow]) < gsz[qrow]
        bs = torch.where(is_top, B[qrow], B[qrow] - 1 - (_h_t(seed, 12, gkey) % 16))
        pid_m = 80000 + _h_t(seed, 13, gkey) % 20001
        f = _h_t(seed, 14, gkey) % 400
        pid_m = torch.where(f < 4, forced[f.clamp(max=3)], pid_m)
        pid_m = torch.where(is_top & share[qrow], q_pid[qrow], pid_m)
        aln = 380 + _h_t(seed, 15, gkey) % 101
        aln = torch.where(is_top & share_aln[qrow], q_aln[qrow], aln)
        acc = (subj * 2654435761 + 12345) % 4294967291                        # order-scrambling, injective on subj
        miss = (_h_t(seed, 16, gkey) % 1000000) < int(p_unmatched * 1000000)
        taxr = torch.where(miss, torch.full_like(subj, 0xFFFFFFFF), subj)
        out.bitscore[r0:r1] = bs.to(torch.int32)
        out.tax_row[r0:r1] = torch.where(taxr >= (1 << 31), taxr - (1 << 32), taxr).to(torch.int32)
        if out.pident is not None:
            out.pident[r0:r1] = pid_t[pid_m]
        if out.pident_milli is not None:
            out.pident_milli[r0:r1] = pid_m.to(torch.int32)
        out.align_len[r0:r1] = aln.to(torch.int32)
        out.acc_rank[r0:r1] = torch.where(acc >= (1 << 31), acc - (1 << 32), acc).to(torch.int32)
    return out


# sum of consensusBeans[].occurrences per result of the reference's zymo-mock golden output: {top-group size: results}
ZYMO_TOP_GROUPS = {1: 30, 2: 390, 3: 381, 4: 18, 5: 4, 6: 526, 7: 352, 8: 2, 9: 554, 10: 10, 15: 3, 16: 2, 17: 3, 26: 7, 28: 1}


def accession_strings(acc_rank_u32: np.ndarray):
    """Accession table whose bytewise order equals the numeric order of acc_rank."""
    uniq, inv = np.unique(acc_rank_u32, return_inverse=True)
    return [f"NR_{int(v):010d}.1" for v in uniq], inv.astype(np.uint32)


CONFIGS = {
    # name: (n_taxa, n_queries, hits_per_query, zipf, deep)
    "C1": dict(n_taxa=2000, n_queries=1000, hits_per_query=10, zipf=None, deep=False),
    "C2": dict(n_taxa=50000, n_queries=100000, hits_per_query=50, zipf=None, deep=False),
    "C3": dict(n_taxa=2400000, n_queries=10000000, hits_per_query=50, zipf=None, deep=False),
    "C5": dict(n_taxa=2400000, n_queries=1000000, hits_per_query=None, zipf=(1.1, 1, 5000), deep=True),
}
