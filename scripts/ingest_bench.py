#!/usr/bin/env python3
"""Ingest throughput: writes a synthetic outfmt-6 table + blutils DB under /tmp and times
blu_ingest_only (text -> SoA columns, no GPU) for a few thread counts.

    python scripts/ingest_bench.py [--queries 400000] [--hits 50] [--taxa 300000] [--threads 1,4,16]
    --accessions clustered  (default) a query's hits come from a neighbourhood of subjects, as BLAST output does
    --accessions uniform    every hit's subject is drawn uniformly (worst case for the accession dictionary)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=400000)
    ap.add_argument("--hits", type=int, default=50)
    ap.add_argument("--taxa", type=int, default=300000)
    ap.add_argument("--threads", default="1,4,16")
    ap.add_argument("--accessions", default="clustered", choices=["clustered", "uniform"])
    ap.add_argument("--dir", default="/tmp/blu_ingest_bench")
    ap.add_argument("--gpu", action="store_true", help="also time the GPU parser (device 0) and check it gives the same columns")
    ap.add_argument("--pipeline", action="store_true", help="also run the whole use-case on the GPU (JSONL out) and print its stage times")
    args = ap.parse_args()
    from blutils_amd import pipeline
    os.makedirs(args.dir, exist_ok=True)
    tj, cache, bt = (os.path.join(args.dir, n) for n in ("tax.blutils.json", "tax.blucache", f"blast.{args.accessions}.tsv"))
    rng = np.random.default_rng(1)
    if not os.path.exists(tj):
        with open(tj, "w") as f:
            f.write('{"blutilsVersion":"8.3.1","sourceDatabase":"synthetic","taxonomies":[')
            for t in range(args.taxa):
                g, fam = t // 12, t // 96
                f.write(("," if t else "") + json.dumps({
                    "taxid": 1000 + t, "rank": "species", "numericLineage": f"d__2;f__{fam};g__{g};s__{1000 + t}",
                    "textLineage": f"d__bacteria;f__fam{fam};g__gen{g};s__sp{t}", "accessions": []}))
            f.write("]}")
    t0 = time.time()
    pipeline.build_db_cache(tj, cache, False)
    print(f"db: {args.taxa} taxids, json {os.path.getsize(tj) / 1e6:.0f} MB, cache built in {time.time() - t0:.2f} s")
    if not os.path.exists(bt):
        with open(bt, "w") as f:
            for q0 in range(0, args.queries, 20000):
                nq = min(20000, args.queries - q0)
                n = nq * args.hits
                if args.accessions == "uniform":
                    sub = rng.integers(0, args.taxa, n)
                else:
                    sub = (np.repeat(rng.integers(0, args.taxa, nq), args.hits) + rng.integers(0, 96, n)) % args.taxa
                pid = rng.integers(80000, 100001, n) / 1000
                aln = rng.integers(380, 480, n)
                bs = rng.integers(200, 2000, n)
                f.write("".join(f"q{q0 + i // args.hits:08d}\tNR_{sub[i]:06d}.1\t{1000 + sub[i]}\t{pid[i]:.3f}\t{aln[i]}\t3\t1\t1\t400\t5\t404\t1e-120\t{bs[i]}\n"
                                for i in range(n)))
    rows = args.queries * args.hits
    size = os.path.getsize(bt)
    print(f"table: {rows} rows, {size / 1e6:.0f} MB of text ({args.accessions} subjects)")
    code = ("import sys, json; sys.path.insert(0, %r); from blutils_amd import pipeline; "
            "print(json.dumps(pipeline.ingest_only(%r, %r)))" % (ROOT, bt, cache))
    ref = None
    for th in [int(x) for x in args.threads.split(",")]:
        env = dict(os.environ, BLU_INGEST_THREADS=str(th), BLU_INGEST_TRACE="1")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True)
        st, ck = json.loads(p.stdout.strip().splitlines()[-1])
        ref = ck if ref is None else ref
        assert ck == ref, "columns differ between thread counts"
        phases = " | ".join(l.split("]")[1].strip() for l in p.stderr.splitlines() if l.startswith("[ingest]"))
        print(f"threads {th:2d}: {st['t_load_hits_s']:.3f} s = {rows / st['t_load_hits_s'] / 1e6:.2f} M rows/s = "
              f"{size / st['t_load_hits_s'] / 1e9:.2f} GB/s   (db load {st['t_load_db_s']:.3f} s)   [{phases}]")
    if args.gpu:
        code_g = ("import sys, json; sys.path.insert(0, %r); from blutils_amd import pipeline; "
                  "pipeline.ingest_only(%r, %r, False, 0); "     # first call: HIP start-up
                  "print(json.dumps(pipeline.ingest_only(%r, %r, False, 0) + (pipeline.last_ingest_path(),)))" % (ROOT, bt, cache, bt, cache))
        env = dict(os.environ, BLU_INGEST_TRACE="1")
        p = subprocess.run([sys.executable, "-c", code_g], env=env, capture_output=True, text=True, check=True)
        st, ck, path = json.loads(p.stdout.strip().splitlines()[-1])
        assert ck == ref and path == "gpu", (ck, ref, path)
        lines = [l.split("]")[1].strip() for l in p.stderr.splitlines() if l.startswith("[ingest-gpu]")]
        phases = " | ".join(lines[len(lines) // 2:])
        print(f"GPU parser: {st['t_load_hits_s']:.3f} s = {rows / st['t_load_hits_s'] / 1e6:.1f} M rows/s = "
              f"{size / st['t_load_hits_s'] / 1e9:.2f} GB/s of text, same columns   [{phases}]")
    if args.pipeline:
        t0 = time.time()
        outp = os.path.join(args.dir, "consensus.jsonl")
        _, st = pipeline.build_consensus_identities(bt, cache, "bacteria", "relaxed", out_format="jsonl", lenient=True, parse=False,
                                                    out_path=outp)
        wall = time.time() - t0
        print(f"pipeline: {wall:.2f} s wall for {st['n_queries']} queries / {st['n_hits']} rows -> {os.path.getsize(outp) / 1e6:.0f} MB of JSONL "
              f"({st['n_queries'] / wall / 1e6:.3f} Mq/s end to end): db {st['t_load_db_s']:.3f} s, ingest {st['t_load_hits_s']:.3f} s, "
              f"engine incl. taxonomy build + PCIe staging {st['t_engine_s']:.3f} s, render {st['t_render_s']:.3f} s")


if __name__ == "__main__":
    main()
