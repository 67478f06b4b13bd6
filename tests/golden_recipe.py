"""Re-synthesise a BLAST hit table from a blutils result's consensus beans.

SURVEY §8(c) reconstruction recipe: the reference's golden output keeps, per
query, the folded consensus beans of the level the scan stopped at.  For each
bean emit `occurrences` rows whose lineage is the bean's taxonomy string,
pident/bit_score those of the result, equal align_length, accession
accessions[k].  Feeding that table through the consensus semantics must give
back (singleMatch, reachedRank, identifier, taxonomy); maxAllowedRank/mutated
come back only when the true reference row was not deeper than the bean's
first-seen lineage (beans keep one lineage per key).
"""
import numpy as np

from oracle import oracle as orc


def table_from_taxa(taxa):
    """taxa: list of golden `taxon` dicts -> oracle HitTable with one query per taxon."""
    seg = [0]
    acc_idx, tax_row, pident, alen, bsc = [], [], [], [], []
    accs, lins = {}, {}
    for t in taxa:
        for bean in t["consensusBeans"]:
            n = int(bean["occurrences"])
            for k in range(n):
                a = bean["accessions"][min(k, len(bean["accessions"]) - 1)]
                acc_idx.append(accs.setdefault(a, len(accs)))
                tax_row.append(lins.setdefault(bean["taxonomy"], len(lins)))
                pident.append(float(t["percIdentity"]))
                alen.append(400)
                bsc.append(int(t["bitScore"]))
        seg.append(len(acc_idx))
    return orc.HitTable(
        seg_off=np.array(seg, dtype=np.uint64),
        acc_idx=np.array(acc_idx, dtype=np.uint32),
        accessions=list(accs.keys()),
        tax_row=np.array(tax_row, dtype=np.int64),
        lineages=list(lins.keys()),
        pident=np.array(pident, dtype=np.float64),
        align_len=np.array(alen, dtype=np.int64),
        bit_score=np.array(bsc, dtype=np.int64),
    )
