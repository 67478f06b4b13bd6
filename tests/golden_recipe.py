"""Re-synthesise a BLAST hit table from a blutils result's consensus beans.

SURVEY §8(c) reconstruction recipe: the reference's golden output keeps, per
query, the folded consensus beans of the level the scan stopped at.  For each
bean emit `occurrences` rows whose lineage is the bean's taxonomy string,
pident/bit_score those of the result, accession accessions[k].  Feeding that
table through the consensus semantics must give back (singleMatch, reachedRank,
identifier, taxonomy); maxAllowedRank/mutated come back only when the true
reference row was not deeper than the bean's first-seen lineage (beans keep one
lineage per key).

align_length: a bean lists its accessions in the order of the reference's sorted
hit list (find_multi_taxa_consensus.rs:39-54: lineage length, perc_identity,
align_length, accession; folded in that order by consensus_result.rs:65-88).
All rows of a bean share lineage and perc_identity here, so the listed order is
the order of (align_length, accession) — and in 585 of the golden's 3586
multi-accession beans it is NOT ascending by accession: the align_lengths
differed.  `ordered=True` (the default) gives the k-th row of a bean
align_length 400 + k, which makes the listed order the one the sort must
produce: the golden then exercises sort keys 3 and 4 and the stable order, and
bean accessions are compared as sequences.  `ordered=False` is the round-1
recipe (every align_length 400).
"""
import numpy as np

from oracle import oracle as orc


def table_from_taxa(taxa, ordered=True, reverse_file_order=False):
    """taxa: list of golden `taxon` dicts -> oracle HitTable with one query per taxon.
    reverse_file_order: the rows of every query in the opposite file order (with `ordered` no two rows of a query tie on
    all four sort keys unless they are the same accession twice, so the result must not depend on the file order)."""
    seg = [0]
    acc_idx, tax_row, pident, alen, bsc = [], [], [], [], []
    accs, lins = {}, {}
    for t in taxa:
        q_first = len(acc_idx)
        for bean in t["consensusBeans"]:
            n = int(bean["occurrences"])
            for k in range(n):
                a = bean["accessions"][min(k, len(bean["accessions"]) - 1)]
                acc_idx.append(accs.setdefault(a, len(accs)))
                tax_row.append(lins.setdefault(bean["taxonomy"], len(lins)))
                pident.append(float(t["percIdentity"]))
                alen.append(400 + k if ordered else 400)
                bsc.append(int(t["bitScore"]))
        if reverse_file_order:
            for col in (acc_idx, tax_row, pident, alen, bsc):
                col[q_first:] = col[q_first:][::-1]
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
