"""Shared helpers of the parity tests: synthetic tables -> oracle inputs, and
rendering of engine records (ints) into the reference's field values (strings)
so that they can be compared with the string-faithful oracle's JSON."""
import numpy as np

from blutils_amd import synth
from oracle import oracle as orc

CUSTOM_16S = {"domain": 50, "kingdom": 60, "phylum": 75, "class": 80, "order": 85, "family": 92, "genus": 97,
              "species": 99}

# engine status <-> string-faithful oracle status
PANIC_OF_STATUS = {16: orc.ST_PANIC_PARSE, 17: orc.ST_PANIC_PARSE, 18: orc.ST_PANIC_ROOT_DISAGREE,
                   19: orc.ST_PANIC_SINGLE_EMPTY}


def oracle_table(tax: synth.SynthTaxonomy, hits: dict, bad=None) -> orc.HitTable:
    """SoA columns (numpy) + synthetic taxonomy -> the row-of-strings table the faithful oracle reads."""
    lineages = tax.lineage_strings()
    if bad is not None:
        for t in np.nonzero(bad)[0]:
            lineages[t] = lineages[t].replace("__", "_", 1)   # an element without `__` fails parse_taxonomy
    accs, acc_idx = synth.accession_strings(hits["acc_rank"].view(np.uint32))
    tr = hits["tax_row"].astype(np.int64)                     # int32 bit pattern: -1 = unmatched
    return orc.HitTable(
        seg_off=hits["seg_off"].astype(np.uint64), acc_idx=acc_idx, accessions=accs,
        tax_row=tr, lineages=lineages, pident=hits["pident"], align_len=hits["align_len"].astype(np.int64),
        bit_score=hits["bitscore"].astype(np.int64))


def columnar(tax: synth.SynthTaxonomy, hits: dict, taxon, strategy, custom=None, bad=None, threads=4):
    return orc.columnar_run(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, hits["seg_off"],
                            hits["bitscore"], hits["tax_row"], hits["pident"], hits["align_len"], hits["acc_rank"],
                            taxon=taxon, strategy=strategy, custom=custom, bad=bad, threads=threads)


class Renderer:
    """Record -> the reference's TaxonomyBean fields, using only the taxonomy arrays and rank tables.
    rank_display/rank_serde: canonical rank code -> string; is_default(tax_row, level) -> bool."""

    def __init__(self, tax: synth.SynthTaxonomy, hits: dict, rank_display, rank_serde, is_default):
        self.tax, self.hits = tax, hits
        self.rank_display, self.rank_serde, self.is_default = rank_display, rank_serde, is_default
        # canonical code of every input rank name: enum kinds 0..8, others from 9 in first-appearance order
        self.canon = []
        others = {}
        enum = {"u": 0, "undefined": 0, "d": 1, "domain": 1, "k": 2, "kingdom": 2, "p": 3, "phylum": 3, "c": 4,
                "class": 4, "o": 5, "order": 5, "f": 6, "family": 6, "g": 7, "genus": 7, "s": 8, "species": 8}
        for name in tax.rank_names:
            low = name.strip().lower()
            if low in enum:
                self.canon.append(enum[low])
            else:
                self.canon.append(others.setdefault(low, 9 + len(others)))

    def render(self, rec) -> dict:
        st = int(rec["status"])
        if st >= 16 or st == 2:
            return {"status": st, "taxon": None}
        row = int(rec["ref_row"])
        t = int(self.hits["tax_row"][row])
        a, b = int(self.tax.lin_off[t]), int(self.tax.lin_off[t + 1])
        nodes = self.tax.lin_node[a:b]
        ranks = [self.canon[r] for r in self.tax.lin_rank[a:b]]
        mask = int(rec["level_mask"])
        levels = [j for j in range(b - a) if (mask >> j) & 1]
        mal = int(rec["max_allowed_level"])
        if mal == 0xFF:
            mar = None
        else:
            code = ranks[mal]
            mar = self.rank_serde(code) if self.is_default(t, mal) else self.rank_display(code)
        return {"status": st, "taxon": {
            "reachedRank": self.rank_serde(int(rec["reached_rank"])),
            "maxAllowedRank": mar,
            "identifier": f"n{int(rec['identifier_node'])}",
            "percIdentity": float(self.hits["pident"][row]),
            "bitScore": float(self.hits["bitscore"][row]),
            "taxonomy": ";".join(f"{self.rank_display(ranks[j])}__n{int(nodes[j])}" for j in levels),
            "mutated": bool(int(rec["flags"]) & 1),
            "singleMatch": st == 1,
        }}


FIELDS = ("reachedRank", "maxAllowedRank", "identifier", "percIdentity", "bitScore", "taxonomy", "mutated",
          "singleMatch")


def assert_matches_faithful(rendered: dict, oracle_json: dict, q=None):
    ost = oracle_json["status"]
    st = rendered["status"]
    if st >= 16:
        assert PANIC_OF_STATUS[st] == ost, (q, st, ost, oracle_json.get("panic"))
        return
    if st == 2:
        assert ost == orc.ST_NO_CONSENSUS, (q, st, ost)
        return
    assert ost == orc.ST_CONSENSUS, (q, st, ost, oracle_json.get("panic"))
    for k in FIELDS:
        assert rendered["taxon"][k] == oracle_json["taxon"][k], (q, k, rendered["taxon"][k], oracle_json["taxon"][k])


def oracle_rank_tables(tax: synth.SynthTaxonomy, taxon, custom):
    """rank_display / rank_serde / is_default built from the ORACLE only (for oracle-vs-oracle tests)."""
    r = Renderer(tax, {}, None, None, None)
    disp, serde = {}, {}
    for name, code in zip(tax.rank_names, r.canon):
        disp[code] = orc.rank_display(name)
        serde[code] = orc.rank_serde(name)
    cache = {}

    def is_default(t, level):
        a, b = int(tax.lin_off[t]), int(tax.lin_off[t + 1])
        key = tuple(tax.lin_rank[a:b])
        if key not in cache:
            cache[key] = orc.interpolate([tax.rank_names[i] for i in key], taxon, custom)[1]
        return bool(cache[key][level])

    return disp.__getitem__, serde.__getitem__, is_default


def adversarial_case(seed, n_tax=40, n_q=3000, max_hits=7):
    """Tiny hand-rolled taxonomy and hit table built to collide on every sort key: few distinct pident /
    align_len / accession values, two bit-score values, lineages that are prefixes of each other, duplicate
    lineages under different taxids, non-default ranks in odd places, segments of 1..max_hits rows."""
    rng = np.random.default_rng(seed)
    rank_names = ["d", "k", "p", "c", "o", "f", "g", "s", "clade", "no rank", "strain", "u", "Domain"]
    pool = [["d", "p", "c", "o", "f", "g", "s"], ["d", "clade", "p", "c", "o", "f", "g", "s"], ["d", "p", "c", "o", "f", "g"],
            ["d", "p", "c", "o", "f", "g", "s", "strain"], ["Domain", "k", "p", "c"], ["d", "p", "c", "o", "f", "no rank", "g", "s"],
            ["d"], ["d", "p"], ["u", "d", "p", "c", "o"], ["d", "p", "clade", "clade", "c"]]
    off, node, rank = [0], [], []
    for t in range(n_tax):
        ranks = pool[int(rng.integers(0, len(pool)))]
        k = int(rng.integers(1, len(ranks) + 1)) if rng.random() < 0.3 else len(ranks)
        parent_choice = int(rng.integers(0, 3))
        for j in range(k):
            # few distinct nodes per level so that lineages share long prefixes; level 0 mostly the same node
            width = 1 if j == 0 and rng.random() < 0.9 else min(1 + j // 2, 3)
            node.append(1000 * j + (parent_choice if j < 3 else int(rng.integers(0, width))))
            rank.append(rank_names.index(ranks[j]))
        off.append(len(node))
    # the ABI wants node ids interned on (Display(rank), identifier): the same identifier under two different
    # canonical ranks is two nodes ("d" and "Domain" are one rank, "u" is another)
    interned = {}
    node = [interned.setdefault((orc.rank_display(rank_names[r]), n), len(interned)) for n, r in zip(node, rank)]
    tax = synth.SynthTaxonomy(rank_names, np.array(off, np.uint64), np.array(node, np.uint32), np.array(rank, np.uint16),
                              (100 + np.arange(n_tax)).astype(np.int64), np.zeros((9, n_tax), np.int32), np.zeros((9, n_tax), np.int32),
                              n_tax, seed, False)
    lens = rng.integers(1, max_hits + 1, n_q)
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    H = int(seg[-1])
    hits = {"seg_off": seg,
            "bitscore": rng.choice([500, 500, 500, 499], H).astype(np.int32),
            "tax_row": rng.integers(0, n_tax, H).astype(np.int32),
            "pident": rng.choice([97.0, 97.0, 99.0, 60.0, 85.0, 45.5], H).astype(np.float64),
            "align_len": rng.choice([400, 400, 401], H).astype(np.int32),
            "acc_rank": rng.integers(0, 4, H).astype(np.int32)}
    hits["tax_row"][rng.random(H) < 0.01] = -1
    return tax, hits
