#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: Mqueries/s of per-query taxonomic consensus + achieved HBM GB/s.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (blu_consensus_run -> one HIP kernel launch) over the whole
synthetic hit table of this rank, inputs already resident in HBM.  Workload at N=1: BASELINE config #3
(10M queries x 50 hits, 2.4M-taxid synthetic taxonomy, relaxed strategy, custom 16S cutoffs).  For N>1
every rank holds its own 10M-query slice (weak scaling: queries are independent, taxonomy replicated,
no data-path collective).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6290 GB/s measured copy
CPU_THREADS_CAP = 16     # worker threads for the CPU legs (the box's CPU share for one GPU)
CUSTOM_16S = {"domain": 50, "kingdom": 60, "phylum": 75, "class": 80, "order": 85, "family": 92, "genus": 97,
              "species": 99}   # reference assets/custom-taxon-cutoffs-bacteria-16S.yaml


LAYOUT_TEXT = {"packed": "bit-score column + 16-byte side records, perc_identity as milli-percent u32 (lossless, 20 B/hit)",
               "milli": "five columns, perc_identity as milli-percent u32 (lossless, 20 B/hit)",
               "f64": "five columns, perc_identity as f64 (24 B/hit)"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--queries", type=int, default=0, help="override the number of queries per GPU")
    ap.add_argument("--taxa", type=int, default=0, help="override the number of taxids")
    ap.add_argument("--hits-per-query", type=int, default=0, help="override the hits per query of C2 / C3 (blutils' own default is max_target_seqs = 10)")
    ap.add_argument("--strategy", default="relaxed", choices=["relaxed", "cautious"])
    ap.add_argument("--taxon", default="custom", choices=["custom", "bacteria", "fungi", "eukaryotes"])
    ap.add_argument("--cpu-sample", type=int, default=500000, help="queries of the workload timed on the CPU oracle")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank holds the full per-GPU workload; strong: the workload is split over the ranks "
                         "(BASELINE config #4: 10M queries sharded across 8 GPUs)")
    ap.add_argument("--pident", default="packed", choices=["packed", "milli", "f64"],
                    help="hit-table layout: packed = bit-score column + 16-byte side records {tax_row, pident_milli, "
                         "align_len, acc_rank} (20 B/hit; a top row's values sit in one memory line); milli = five "
                         "columns with perc_identity as milli-percent u32 (20 B/hit); f64 = five columns with "
                         "perc_identity as f64 (24 B/hit, the canonical layout of BASELINE.md)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one run (both kernels) in a HIP graph and time replays: for launch-bound sizes (C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from blutils_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1 or os.environ.get("BLU_BENCH_FORCE_DIST") == "1"   # (the override lets a 1-GPU box rehearse the RCCL path)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device(dev))

    cfg = dict(synth.CONFIGS[args.config])
    if args.queries:
        cfg["n_queries"] = args.queries
    if args.taxa:
        cfg["n_taxa"] = args.taxa
    if args.hits_per_query and cfg["zipf"] is None:
        cfg["hits_per_query"] = args.hits_per_query
    if args.scaling == "strong" and world > 1:
        cfg["n_queries"] = (cfg["n_queries"] + world - 1) // world      # per-GPU query slice of one fixed table
    seed = synth.SEEDS[args.config]
    custom = CUSTOM_16S if args.taxon == "custom" else None

    t0 = time.time()
    tax = synth.make_taxonomy(cfg["n_taxa"], seed, deep=cfg["deep"])
    t_tax = time.time() - t0
    t0 = time.time()
    eng_tax = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon=args.taxon,
                              custom=custom, device=local_rank)
    t_up = time.time() - t0
    t0 = time.time()
    hits = synth.make_hits(tax, cfg["n_queries"], seed, cfg["hits_per_query"], zipf=cfg["zipf"], device=dev,
                           q_offset=rank * cfg["n_queries"], columns="f64" if args.pident == "f64" else "milli")
    torch.cuda.synchronize()
    t_hits = time.time() - t0
    Q, Hn = hits.n_queries, hits.n_hits
    out = torch.zeros(32 * Q, dtype=torch.uint8, device=dev)
    hd = hits.as_dict("f64" if args.pident == "f64" else "milli")
    # the join of the hit table with the taxonomy (mod.rs:72-76): desc row -> engine row id, done once at ingest.
    # The oracle legs read the desc rows of the sampled prefix, kept aside.
    S_keep = min(Q, max(args.cpu_sample, 1))
    desc_rows_sample = hd["tax_row"][: int(hits.seg_off[S_keep].item())].cpu().numpy()
    for a in range(0, Hn, 1 << 26):
        b = min(Hn, a + (1 << 26))
        hd["tax_row"][a:b] = eng_tax.engine_rows(hd["tax_row"][a:b])
    cols = hd            # the five columns (kept for the oracle sample below)
    if args.pident == "packed":
        hits.tax_row = hd["tax_row"]
        hd = hits.as_dict("packed")
    if rank == 0:
        log(f"[bench] taxonomy {tax.n} taxids ({t_tax:.1f}s gen, {t_up:.1f}s upload, {eng_tax.n_shapes} shapes, "
            f"depth<={eng_tax.max_depth}, {eng_tax.device_bytes / 1e6:.0f} MB on device); "
            f"hits {Q} queries / {Hn} rows ({t_hits:.1f}s gen on GPU)")

    def step():
        engine.run_consensus_device(eng_tax, hd, out, strategy=args.strategy)

    # ---- parity gate (rank 0): GPU records of a sample == columnar oracle, before any timing is accepted
    cpu_baseline = None
    if rank == 0 and not args.no_parity_gate:
        from oracle import oracle as orc
        step()
        torch.cuda.synchronize()
        S = min(Q, max(args.cpu_sample, 1))
        seg = hits.seg_off[: S + 1].cpu().numpy()
        nrow = int(seg[-1])
        samp = {k: v[:nrow].cpu().numpy() for k, v in cols.items() if k not in ("seg_off", "tax_row")}
        samp["tax_row"] = desc_rows_sample[:nrow]
        if args.pident != "f64":   # the oracle reads the f64 the reference's parser would produce: k / 1000, correctly rounded
            samp["pident"] = samp.pop("pident_milli").astype(np.float64) / 1000.0
        got = engine.records_from_tensor(out[: 32 * S])
        exp = orc.columnar_run(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, seg, samp["bitscore"],
                               samp["tax_row"], samp["pident"], samp["align_len"], samp["acc_rank"],
                               taxon=args.taxon, strategy=args.strategy, custom=custom, threads=CPU_THREADS_CAP)
        if got.tobytes() != exp.tobytes():
            bad = np.nonzero(got.view(np.uint8).reshape(-1, 32) != exp.view(np.uint8).reshape(-1, 32))[0]
            raise SystemExit(f"parity gate FAILED: {len(np.unique(bad))} of {S} sampled queries differ from the oracle")
        log(f"[bench] parity gate ok: {S} queries bit-identical to the oracle")
        if not args.no_cpu_baseline and world == 1:
            # host threads this process may use; the GPU box gives one GPU's share (16) of a 256-thread host
            cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), CPU_THREADS_CAP)
            dt, run = orc.faithful_on_synthetic(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, seg,
                                                samp["bitscore"], samp["tax_row"], samp["pident"], samp["align_len"],
                                                samp["acc_rank"], taxon=args.taxon, strategy=args.strategy,
                                                custom=custom, threads=cores)
            run.close()
            cpu_baseline = {"value": S / dt / 1e6, "unit": "Mqueries/s", "cores": cores, "kind": "port",
                            "sample": f"first {S} queries ({nrow} hit rows) of the same table, string-faithful C++ "
                                      f"restatement of the Rust path (oracle/blu_oracle.cpp), {cores} threads over "
                                      f"queries, {dt:.2f} s wall"}
            log(f"[bench] cpu baseline: {cpu_baseline['value']:.4f} Mq/s on {cores} threads ({dt:.2f}s)")

    if args.graph:
        # the run leaves its worklist counters as it found them, so the captured pair of kernels can be replayed
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()                                   # workspace allocation happens outside the capture
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
        step = graph.replay

    # ---- timed region
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for a, b in ev:
        a.record()          # torch's current stream == the stream the kernel is launched on
        step()
        b.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        total_q = Q * world
        value = total_q * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        alg_bytes = hits.algorithmic_bytes(args.pident)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile) and not (args.queries or args.taxa or args.hits_per_query):
            try:
                tkey = args.config if args.pident == "f64" else args.config + "-" + args.pident
                traffic = json.load(open(tfile)).get(tkey, {}).get("traffic_bytes_per_launch")
            except Exception:
                traffic = None
        # the per-unit byte formula, but "never a larger figure than what is physically read" — the stream
        # kernel skips the lines of the four non-bit-score columns that hold no top row, so on C3 the PMC-measured
        # traffic of a launch (same seeded table) is BELOW the formula; the smaller of the two prices the roofline
        used_bytes = min(alg_bytes, traffic) if traffic else alg_bytes
        achieved = used_bytes / (k_ms * 1e-3) / 1e9
        name, grid, block = engine.last_launch()
        line = {
            "metric": "Mqueries/sec consensus (synthetic outfmt-6 hit table)",
            "value": value, "unit": "Mqueries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "i32+f64", "data": "synthetic",
            "config": {"workload": f"{args.config}: {Q} queries x "
                                   f"{cfg['hits_per_query'] if cfg['zipf'] is None else 'Zipf' + str(cfg['zipf'])} hits per GPU, "
                                   f"{tax.n}-taxid synthetic taxonomy, strategy {args.strategy}, taxon {args.taxon}, "
                                   f"layout {LAYOUT_TEXT[args.pident]}",
                       "queries_per_gpu": Q, "hit_rows_per_gpu": Hn, "taxids": tax.n, "strategy": args.strategy,
                       "taxon": args.taxon, "pident_layout": args.pident, "bytes_per_hit": 24 if args.pident == "f64" else 20, "seed": hex(seed), "generator_version": synth.GENERATOR_VERSION,
                       "parallelism": f"query-sharded x{world}, taxonomy replicated, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": name, "kernel_ms": k_ms, "algorithmic_bytes": alg_bytes, "bytes_used": used_bytes,
                         "by_formula": {"achieved": alg_bytes / (k_ms * 1e-3) / 1e9, "frac": alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                         "launch": {"grid": grid, "block": block}},
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
