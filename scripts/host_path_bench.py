#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer boundary (blu_consensus_run with on_device = 0): the library allocates,
stages the columns over PCIe, runs and copies the records back.  Never the bench `value`; quoted in DESIGN.md."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=2000000)
    ap.add_argument("--taxa", type=int, default=300000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from blutils_amd import engine, synth
    tax = synth.make_taxonomy(args.taxa, 11)
    dh = synth.make_hits(tax, args.queries, 11, 50, device="cuda")
    t = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon="bacteria", device=0, taxid=tax.taxid)
    h = dh.numpy()
    rows = t.engine_rows(h["tax_row"])
    milli = dh.pident_milli.cpu().numpy()
    nbytes = 20 * len(rows) + 8 * (args.queries + 1) + 32 * args.queries
    for name, kw in (("milli", dict(pident=None, pident_milli=milli)), ("f64", dict(pident=h["pident"]))):
        best = 1e9
        for _ in range(args.reps):
            t0 = time.perf_counter()
            engine.run_consensus_host(t, h["seg_off"], h["bitscore"], rows, kw.get("pident"), h["align_len"], h["acc_rank"], "relaxed",
                                      pident_milli=kw.get("pident_milli"))
            best = min(best, time.perf_counter() - t0)
        b = nbytes + (4 * len(rows) if name == "f64" else 0)
        print(f"{name}: {args.queries} queries / {len(rows)} rows from pageable host memory: {best * 1e3:.1f} ms = "
              f"{args.queries / best / 1e6:.1f} Mq/s, {b / best / 1e9:.1f} GB/s over the boundary")


if __name__ == "__main__":
    main()
