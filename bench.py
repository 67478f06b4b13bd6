#!/usr/bin/env python3
"""bench.py — BASELINE.json metric: Mqueries/s of per-query taxonomic consensus + achieved HBM GB/s.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (blu_consensus_run -> the HIP kernels) over the whole synthetic hit table of this
rank, inputs already resident in HBM.

* N = 1: BASELINE config 3 (10 M queries x 50 hits, 2.4 M-taxid synthetic taxonomy, relaxed strategy, custom 16S cutoffs).
* N > 1: BASELINE config 4 — the SAME 10 M-query table cut into N contiguous query ranges balanced by hit count
  (blutils_amd/shard.py), one range per rank, taxonomy replicated, no data-path collective ("scaling": "strong").  The
  weak-scaling figure (every rank a 10 M-query table of its own) is measured in the same run and reported as the
  secondary field `weak_scaling`.  `--scaling weak` makes it the headline instead.

Rank 0 prints ONE JSON line.  `roofline`:
  useful_bytes   bytes the reference semantics need from this table: every bit-score (4 H), the other 16 B (20 B in the
                 f64 layout) of the T top-score rows only, the offsets 8 (Q + 1) and the records 32 Q; T is counted on
                 the device from the timed table.  `achieved` / `frac` are priced with these.
  by_formula     SURVEY 8d's per-unit figure, 20 (or 24) B per hit row: an upper bound on what a kernel may read.
  traffic        HBM bytes per launch from the PMC passes of profiles/ (profiles/hbm_traffic.json), quoted only when
                 that file was taken with the kernel source of this tree (sha256 match) and this workload; else null.
"""
import argparse
import hashlib
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
KERNEL_SOURCES = ("blutils_amd/csrc/consensus_kernel.hip", "blutils_amd/csrc/blu_internal.h")


LAYOUT_TEXT = {"packed": "bit-score column + 16-byte side records, perc_identity as milli-percent u32 (lossless, 20 B/hit)",
               "packed64": "bit-score column + 24-byte side records with perc_identity as f64 (28 B/hit)",
               "milli": "five columns, perc_identity as milli-percent u32 (lossless, 20 B/hit)",
               "f64": "five columns, perc_identity as f64 (24 B/hit)"}


# the arithmetic the path computes in, per layout: integer compares on bit-scores (i32) and milli-percent identities / packed
# sort keys (u32), one f64 conversion per record; the f64 layouts compare f64 identities where they are off the milli-percent grid
DTYPE_TEXT = {"packed": "i32/u32", "milli": "i32/u32", "packed64": "i32/u32+f64", "f64": "i32/u32+f64"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def kernel_source_sha256() -> str:
    """Identity of the kernel the counters in profiles/hbm_traffic.json belong to."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def count_top_rows(hits) -> int:
    """T = rows that tie on their query's top bit-score (all the reference ever parses beyond the score column:
    find_single_query_consensus.rs:51-64), counted on the device."""
    import torch
    seg = hits.seg_off
    Q = hits.n_queries
    total = 0
    step = 1 << 20
    for q0 in range(0, Q, step):
        q1 = min(Q, q0 + step)
        r0, r1 = int(seg[q0].item()), int(seg[q1].item())
        if r1 == r0:
            continue
        lens = seg[q0 + 1:q1 + 1] - seg[q0:q1]
        qid = torch.repeat_interleave(torch.arange(q1 - q0, device=seg.device), lens)
        bs = hits.bitscore[r0:r1]
        top = torch.full((q1 - q0,), -(1 << 31), dtype=torch.int32, device=seg.device)
        top.scatter_reduce_(0, qid, bs, reduce="amax", include_self=True)
        total += int((bs == top[qid]).sum().item())
    return total


def useful_bytes(pident: str, n_hits: int, n_queries: int, top_rows: int) -> int:
    """Bytes the reference semantics need from a table in this layout: every bit-score, the other values of the top-score
    rows only (16 B in the milli-percent layouts, 20 B with an f64 column, 24 B as f64 side records), offsets, records."""
    side = {"f64": 20, "packed64": 24}.get(pident, 16)
    return 4 * n_hits + side * top_rows + 8 * (n_queries + 1) + 32 * n_queries


def build_hit_table(synth, engine, torch, tax, eng_tax, cfg, seed, n_queries, q_offset, top_group, pident, dev, sample):
    """Synthetic table of `n_queries` queries resident on `dev` in layout `pident`, joined with the taxonomy
    (desc row -> engine row id: the left join of mod.rs:72-76, done once at ingest).  Returns (hits, device dict the
    engine reads, the five columns, desc rows of the first `sample` queries for the oracle legs, seconds spent generating)."""
    f64_cols = pident in ("f64", "packed64")
    t0 = time.time()
    hits = synth.make_hits(tax, n_queries, seed, cfg["hits_per_query"], zipf=cfg["zipf"], device=dev, q_offset=q_offset,
                           columns="f64" if f64_cols else "milli", top_group=top_group)
    torch.cuda.synchronize()
    t_hits = time.time() - t0
    hd = hits.as_dict("f64" if f64_cols else "milli")
    S_keep = min(hits.n_queries, max(sample, 1))
    desc_rows_sample = hd["tax_row"][: int(hits.seg_off[S_keep].item())].cpu().numpy()
    for a in range(0, hits.n_hits, 1 << 26):
        b = min(hits.n_hits, a + (1 << 26))
        hd["tax_row"][a:b] = eng_tax.engine_rows(hd["tax_row"][a:b])
    cols = hd            # the five columns (kept for the oracle sample)
    if pident in ("packed", "packed64"):
        hits.tax_row = hd["tax_row"]
        # blu_hits_pack / blu_hits_pack64: the side records, with the shape hints (BLU_BENCH_NO_HINTS=1, experiments: records put
        # together by hand, no hints)
        hd = hits.as_dict(pident, tax=None if (os.environ.get("BLU_BENCH_NO_HINTS") == "1" and pident == "packed") else eng_tax)
    return hits, hd, cols, desc_rows_sample, t_hits


def oracle_sample(np, hits, cols, desc_rows_sample, S, f64_cols):
    """The first S queries of the table as the numpy columns the oracles read."""
    seg = hits.seg_off[: S + 1].cpu().numpy()
    nrow = int(seg[-1])
    samp = {k: v[:nrow].cpu().numpy() for k, v in cols.items() if k not in ("seg_off", "tax_row")}
    samp["tax_row"] = desc_rows_sample[:nrow]
    if not f64_cols:   # the oracle reads the f64 the reference's parser would produce: k / 1000, correctly rounded
        samp["pident"] = samp.pop("pident_milli").astype(np.float64) / 1000.0
    return seg, nrow, samp


# Workloads measured after the headline in the same run (N = 1): what DESIGN's table quotes, on the driver's box.
SECONDARY = (
    dict(name="C3, top groups as in the reference's real output (zymo-mock histogram, mean 5.7 rows)", config="C3", top_group="zymo", pident="packed"),
    dict(name="C3 shape, all 50 hits of every query tied (table read in full), 2 M queries", config="C3", top_group="all", queries=2000000, pident="packed"),
    dict(name="C3 in the canonical f64 layout (five columns, 24 B/hit)", config="C3", pident="f64"),
    dict(name="C3 with f64 side records (blu_hits_pack64, 28 B/hit)", config="C3", pident="packed64"),
    dict(name="C3 as five columns with perc_identity as milli-percent u32 (20 B/hit)", config="C3", pident="milli"),
    dict(name="C4 slice: one eighth of C3 (1.25 M queries), what one rank of the 8-GPU run holds", config="C3", queries=1250000, pident="packed"),
    dict(name="C5: 1 M queries, Zipf 1..5000 hits, deep lineages", config="C5", pident="packed"),
    dict(name="C2: 100 k queries x 50 hits, 50 k taxids, replayed from a HIP graph", config="C2", pident="packed", graph=True),
)


def stream_read_ceiling(torch, gib: float = 4.0):
    """Read-only HBM streaming rate of THIS box in GB/s (scripts/probe/stream_probe.hip, built by __graft_entry__.build() as
    blutils_amd/lib/libblu_probe.so): the ceiling SURVEY 8d asks to quote next to the 8 TB/s spec peak.  None if the probe
    library is not there."""
    import ctypes
    so = os.path.join(ROOT, "blutils_amd", "lib", "libblu_probe.so")
    if not os.path.exists(so):
        return None
    L = ctypes.CDLL(so)
    L.probe_read.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    nbytes = int(gib * (1 << 30))
    buf = torch.empty(nbytes // 4, dtype=torch.int32, device="cuda").fill_(1)
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    best = 0.0
    shapes = [(L.probe_read, g) for g in (2048, 4096, 8192)]
    if hasattr(L, "probe_read_plain"):           # (512-thread blocks, one load per thread and step)
        L.probe_read_plain.argtypes = L.probe_read.argtypes
        shapes += [(L.probe_read_plain, g) for g in (1024, 2048, 4096)]
    for fn, grid in shapes:
        for _ in range(2):
            fn(buf.data_ptr(), nbytes, sink.data_ptr(), grid, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn(buf.data_ptr(), nbytes, sink.data_ptr(), grid, s)
        b.record()
        torch.cuda.synchronize()
        rate = nbytes * 5 / (a.elapsed_time(b) * 1e-3) / 1e9
        if os.environ.get("BLU_BENCH_CEILING_TRACE"):
            log(f"[bench]   ceiling probe {fn.__name__} grid {grid}: {rate:.0f} GB/s")
        best = max(best, rate)
    del buf
    torch.cuda.empty_cache()
    return best


def time_pack(engine, torch, np, eng_tax, cols, wide: bool, n_hits: int, reps: int = 5):
    """blu_hits_pack / blu_hits_pack64 on the timed table's columns, HIP events on the launch stream: what a caller that
    holds SoA columns pays once per table before the packed layout's runs.  Bytes: the four 4-byte columns (f64 layout:
    8 for perc_identity) read + the record written; the 2-byte shape hint per row comes from a table that stays in L2."""
    outp = engine.pack_hits_device(eng_tax, cols, wide=wide)      # (warm-up; allocates the records)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        outp = engine.pack_hits_device(eng_tax, cols, wide=wide)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    del outp
    torch.cuda.empty_cache()
    nbytes = ((20 + 24) if wide else (16 + 16)) * n_hits
    return {"pack_ms": ms, "bytes": nbytes, "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "unit": "GB/s", "note": "blu_hits_pack on this table's columns (columns read + records written per hit row), once per table; "
                                    "not inside ms_per_step — the GPU parser and the pipeline emit the packed layout directly"}


def end_to_end_entry(queries: int = 2000000):
    """The whole use-case in a fresh process per repetition (scripts/e2e_bench.py: outfmt-6 text + taxonomy cache in, JSONL file
    out; HIP start-up, the PCIe upload of the text, GPU parse, consensus, rendering and the file write all inside the wall time
    of the call) as a `secondary` entry.  None — never an exception — when the box cannot run it (no gcc, no room in /tmp)."""
    import shutil
    import subprocess
    try:
        d = "/tmp/blu_e2e_bench"
        if shutil.disk_usage("/tmp").free < 12 * (1 << 30):
            return None
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "e2e_bench.py"), "--queries", str(queries), "--reps", "3", "--dir", d],
                           capture_output=True, text=True, timeout=240)
        shutil.rmtree(d, ignore_errors=True)
        if p.returncode != 0:
            log("[bench] end-to-end entry failed: " + p.stderr[-300:])
            return None
        r = json.loads(p.stdout.strip().splitlines()[-1])
        log(f"[bench] secondary: end to end, {r['queries']} queries / {r['text_gb']:.1f} GB of text: {r['wall_s']:.3f} s = {r['e2e_mqps']:.2f} Mq/s ({time.time() - t0:.0f} s)")
        return {"workload": "end to end: outfmt-6 text + taxonomy cache -> JSONL file, fresh process (HIP start-up, PCIe upload, GPU parse, "
                            "consensus, render, write inside the wall time; scripts/e2e_bench.py)", "config": "C3 shape", "queries": r["queries"],
                "hit_rows": r["rows"], "text_GB": r["text_gb"], "wall_s": r["wall_s"], "wall_s_all": r.get("wall_s_all"), "value": r["e2e_mqps"], "unit": "Mqueries/s",
                "note": "best of three fresh processes (all three in wall_s_all: HIP start-up alone varies by 0.05-0.25 s from one process to the next); "
                        "PCIe- and start-up-bound: never the headline value"}
    except Exception as e:   # the headline must not depend on this
        log(f"[bench] end-to-end entry skipped: {e}")
        return None


def run_secondary(args, synth, engine, torch, np, dev, local_rank, custom, reuse):
    """One entry per SECONDARY workload: kernel_ms (HIP events on the launch stream, mean of `steps`), ms_per_step (wall),
    value (Mq/s by the wall clock), useful bytes and the roofline fraction they give; a 20 000-query parity gate each."""
    out_list = []
    tax_cache = dict(reuse)                      # config name -> (tax, eng_tax)
    for w in SECONDARY:
        t_begin = time.time()
        cfg = dict(synth.CONFIGS[w["config"]])
        seed = synth.SEEDS[w["config"]]
        if w["config"] not in tax_cache:
            tx = synth.make_taxonomy(cfg["n_taxa"], seed, deep=cfg["deep"])
            tax_cache[w["config"]] = (tx, engine.Taxonomy(tx.lin_off, tx.lin_node, tx.lin_rank, tx.rank_names, taxon=args.taxon,
                                                          custom=custom, device=local_rank))
        tax, eng_tax = tax_cache[w["config"]]
        nq = w.get("queries", cfg["n_queries"])
        S = min(nq, 20000)
        hits, hd, cols, desc, _ = build_hit_table(synth, engine, torch, tax, eng_tax, cfg, seed, nq, 0, w.get("top_group", "geo"),
                                                  w["pident"], dev, S)
        out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device=dev)
        step = lambda: engine.run_consensus_device(eng_tax, hd, out, strategy=args.strategy)
        entry = {"workload": w["name"], "config": w["config"], "pident_layout": w["pident"], "queries": hits.n_queries, "hit_rows": hits.n_hits}
        fn = step
        if w.get("graph"):
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                step()
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                step()
            fn = graph.replay
        for _ in range(min(args.warmup, 3)):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a, b in ev:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        k_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        if not args.no_parity_gate:
            from oracle import oracle as orc
            torch.cuda.synchronize()           # (the records of the last timed step)
            seg, nrow, samp = oracle_sample(np, hits, cols, desc, S, w["pident"] in ("f64", "packed64"))
            exp = orc.columnar_run(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, seg, samp["bitscore"], samp["tax_row"],
                                   samp["pident"], samp["align_len"], samp["acc_rank"], taxon=args.taxon, strategy=args.strategy,
                                   custom=custom, threads=CPU_THREADS_CAP)
            if engine.records_from_tensor(out[: 32 * S]).tobytes() != exp.tobytes():
                raise SystemExit(f"parity gate FAILED on the secondary workload {w['name']!r}")
            entry["parity_gate_queries"] = S
        T_top = count_top_rows(hits)
        useful = useful_bytes(w["pident"], hits.n_hits, hits.n_queries, T_top)
        entry.update({"kernel_ms": k_ms, "ms_per_step": wall * 1e3 / args.steps, "value": hits.n_queries * args.steps / wall / 1e6,
                      "unit": "Mqueries/s", "useful_bytes": useful, "top_rows": T_top,
                      "achieved": useful / (k_ms * 1e-3) / 1e9, "frac": useful / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                      "seconds": None})
        del hits, hd, cols, out
        torch.cuda.empty_cache()
        entry["seconds"] = round(time.time() - t_begin, 1)
        log(f"[bench] secondary: {w['name']}: {entry['kernel_ms']:.4f} ms, {entry['value']:.0f} Mq/s, frac {entry['frac']:.3f} ({entry['seconds']} s)")
        out_list.append(entry)
    return out_list


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, the same argv), rank 0's stdout is this process's stdout.  Returns the worst exit code."""
    import socket
    import subprocess
    share = os.environ.get("BLU_BENCH_SHARE_GPU") == "1"
    try:
        import torch
        have = torch.cuda.device_count()            # (counting devices does not initialise the GPU on this image)
    except Exception:
        have = 0
    if have < n and not share:
        log(f"[bench] --gpus {n}: this node shows {have} GPU(s)")
        return 2
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad:                                      # a rank that died leaves the others in a barrier: end them (by PID)
            rc = abs(bad[0])
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C2", "C3", "C5"])
    ap.add_argument("--queries", type=int, default=0, help="override the number of queries of the table")
    ap.add_argument("--taxa", type=int, default=0, help="override the number of taxids")
    ap.add_argument("--hits-per-query", type=int, default=0, help="override the hits per query of C2 / C3 (blutils' own default is max_target_seqs = 10)")
    ap.add_argument("--top-group", default="geo", choices=["geo", "zymo", "all"],
                    help="size of the top bit-score group: geo = 1 + Geometric(0.35) (SURVEY 8d, the headline); zymo = histogram of "
                         "the reference's real zymo-mock output (mean 5.7); all = every hit ties (table read in full)")
    ap.add_argument("--strategy", default="relaxed", choices=["relaxed", "cautious"])
    ap.add_argument("--taxon", default="custom", choices=["custom", "bacteria", "fungi", "eukaryotes"])
    ap.add_argument("--cpu-sample", type=int, default=500000, help="queries of the workload timed on the CPU oracle")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="auto: strong for N > 1 (BASELINE config 4: ONE table sharded over the ranks), with the weak figure as a "
                         "secondary field; weak: every rank holds a full table of its own")
    ap.add_argument("--pident", default="packed", choices=["packed", "milli", "f64", "packed64"],
                    help="hit-table layout: packed = bit-score column + 16-byte side records {tax_row, pident_milli, "
                         "align_len, acc_rank} (20 B/hit; a top row's values sit in one memory line); milli = five "
                         "columns with perc_identity as milli-percent u32 (20 B/hit); f64 = five columns with "
                         "perc_identity as f64 (24 B/hit, the canonical layout of BASELINE.md); packed64 = bit-score column + "
                         "24-byte side records {tax_row, hint, align_len, acc_rank, pident f64} (28 B/hit: any f64 identity, a "
                         "top row's values in one or two memory lines instead of five)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one run (both kernels) in a HIP graph and time replays: for launch-bound sizes (C2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements (N = 1: the other workloads of SECONDARY; N > 1: the weak-scaling figure)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Launched without a launcher: start the N rank processes here.  This parent makes no GPU call of any kind before
        # (or after) it spawns them — fresh children, no re-exec of a process that has touched the card.
        raise SystemExit(self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("BLU_BENCH_FORCE_DIST") != "1":
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself; or torch.distributed.run --nproc-per-node N)")

    import numpy as np
    import torch
    import torch.distributed as dist

    from blutils_amd import engine, shard, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force_dist = os.environ.get("BLU_BENCH_FORCE_DIST") == "1"   # lets a 1-GPU box rehearse the RCCL path and the N > 1 defaults
    # BLU_BENCH_SHARE_GPU=1: a rehearsal of the N > 1 logic (one table cut over the ranks, max-over-ranks timing) on a box with
    # ONE GPU — every rank uses cuda:0 and the ranks meet over gloo (RCCL refuses two ranks on one device).  Its `value` is
    # not a scaling measurement: the ranks share the card.
    share_gpu = os.environ.get("BLU_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    distributed = world > 1 or force_dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    cdev = "cpu" if share_gpu else dev               # where the few scalars of the collectives live
    backend = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if share_gpu:
            dist.init_process_group(backend="gloo")
            backend = f"gloo (rehearsal: {dist.get_world_size()} ranks share one GPU), world_size {dist.get_world_size()}"
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))
            backend = f"{dist.get_backend()} (RCCL), world_size {dist.get_world_size()}"
    scaling = args.scaling if args.scaling != "auto" else ("strong" if distributed else "weak")

    cfg = dict(synth.CONFIGS[args.config])
    if args.queries:
        cfg["n_queries"] = args.queries
    if args.taxa:
        cfg["n_taxa"] = args.taxa
    if args.hits_per_query and cfg["zipf"] is None:
        cfg["hits_per_query"] = args.hits_per_query
    seed = synth.SEEDS[args.config]
    custom = CUSTOM_16S if args.taxon == "custom" else None
    Q_table = cfg["n_queries"]                       # queries of ONE table (strong: cut over the ranks; weak: per rank)

    t0 = time.time()
    tax = synth.make_taxonomy(cfg["n_taxa"], seed, deep=cfg["deep"])
    t_tax = time.time() - t0
    t0 = time.time()
    eng_tax = engine.Taxonomy(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, taxon=args.taxon,
                              custom=custom, device=local_rank)
    t_up = time.time() - t0

    f64_cols = args.pident in ("f64", "packed64")   # layouts whose perc_identity is an f64

    def query_range(mode):
        """[q0, q1) of this rank in the global query numbering."""
        if mode == "weak" or world == 1:
            return rank * Q_table, (rank + 1) * Q_table
        if cfg["zipf"] is None:                      # hit-balanced contiguous ranges of the one table (shard.py)
            seg = np.arange(Q_table + 1, dtype=np.int64) * cfg["hits_per_query"]
        else:                                        # the skewed table: its hit counts come from the generator
            seg = np.concatenate([[0], np.cumsum(synth.hit_counts(Q_table, seed, None, cfg["zipf"], dev).cpu().numpy())])
        return shard.balanced_query_ranges(seg, world)[rank]

    def build_table(mode):
        q0, q1 = query_range(mode)
        hits, hd, cols, desc_rows_sample, t_hits = build_hit_table(synth, engine, torch, tax, eng_tax, cfg, seed, q1 - q0, q0, args.top_group,
                                                                   args.pident, dev, args.cpu_sample)
        out = torch.zeros(32 * hits.n_queries, dtype=torch.uint8, device=dev)
        return hits, hd, cols, desc_rows_sample, out, t_hits

    hits, hd, cols, desc_rows_sample, out, t_hits = build_table(scaling)
    Q, Hn = hits.n_queries, hits.n_hits
    if rank == 0:
        log(f"[bench] taxonomy {tax.n} taxids ({t_tax:.1f}s gen, {t_up:.1f}s upload, {eng_tax.n_shapes} shapes, "
            f"depth<={eng_tax.max_depth}, {eng_tax.device_bytes / 1e6:.0f} MB on device); "
            f"hits {Q} queries / {Hn} rows on this rank ({t_hits:.1f}s gen on GPU); scaling {scaling}, world {world}")

    state = {"hd": hd, "out": out}
    if rank == 0 and os.environ.get("BLU_BENCH_PTRS"):    # (placement diagnostics: scripts/calls/README.md, call 49)
        log("[bench] ptrs " + " ".join(f"{k}={v.data_ptr():#x}/{v.numel() * v.element_size() >> 20}MiB" for k, v in hd.items() if hasattr(v, "data_ptr"))
            + f" out={out.data_ptr():#x}")

    def step():
        engine.run_consensus_device(eng_tax, state["hd"], state["out"], strategy=args.strategy)

    run_step = step
    if args.graph:
        # the run leaves its worklist counters as it found them, so the captured pair of kernels can be replayed
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            step()                                   # workspace allocation happens outside the capture
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
        run_step = graph.replay

    def timed(fn):
        """W untimed + K timed steps between barriers; (wall seconds, max over ranks; this rank's per-step event times)."""
        for _ in range(args.warmup):
            fn()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t_start = time.perf_counter()
        for a, b in ev:
            a.record()          # torch's current stream == the stream the kernels are launched on
            fn()
            b.record()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        kernel_ms = [a.elapsed_time(b) for a, b in ev]
        if distributed:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, kernel_ms

    # ---- timed region (the headline)
    elapsed, kernel_ms = timed(run_step)

    # ---- parity gate (rank 0): the records the LAST TIMED STEP wrote, for a sample of the queries, == the columnar oracle.
    # No line is printed unless it passes.  It runs after the timed region, not before it: the oracle legs keep the GPU
    # idle for a second or two, and a timed region that starts on a card that has just been idle measures its clock ramp
    # (the same build: 1.07 ms behind the gate, 1.03 ms without it, one box).
    cpu_baseline = None
    if rank == 0 and not args.no_parity_gate:
        from oracle import oracle as orc
        torch.cuda.synchronize()
        S = min(Q, max(args.cpu_sample, 1))
        seg, nrow, samp = oracle_sample(np, hits, cols, desc_rows_sample, S, f64_cols)
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
            # the non-allocating CPU ceiling — the columnar restatement on the same SoA sample, same threads
            t0 = time.perf_counter()
            orc.columnar_run(tax.lin_off, tax.lin_node, tax.lin_rank, tax.rank_names, seg, samp["bitscore"], samp["tax_row"], samp["pident"],
                             samp["align_len"], samp["acc_rank"], taxon=args.taxon, strategy=args.strategy, custom=custom, threads=cores)
            dtc = time.perf_counter() - t0
            cpu_baseline["columnar"] = {"value": S / dtc / 1e6, "unit": "Mqueries/s", "cores": cores, "kind": "port",
                                        "sample": f"the same {S} queries through the columnar restatement (oracle/blu_oracle_columnar.cpp: interned SoA "
                                                  f"input, no strings, incl. its per-call table set-up), {dtc:.2f} s wall"}
            log(f"[bench] cpu baseline (columnar): {cpu_baseline['columnar']['value']:.3f} Mq/s on {cores} threads ({dtc:.2f}s)")

    pack_info = None
    if rank == 0 and world == 1 and args.pident in ("packed", "packed64") and not args.no_secondary:
        pc = {k: cols[k] for k in ("tax_row", "align_len", "acc_rank")}
        pc["pident" if args.pident == "packed64" else "pident_milli"] = cols["pident" if args.pident == "packed64" else "pident_milli"]
        pack_info = time_pack(engine, torch, np, eng_tax, pc, args.pident == "packed64", Hn)
        log(f"[bench] pack: {pack_info['pack_ms']:.3f} ms for {Hn} rows ({pack_info['achieved']:.0f} GB/s)")

    total_q = Q
    if distributed:
        tq = torch.tensor([Q], dtype=torch.int64, device=cdev)
        dist.all_reduce(tq)
        total_q = int(tq.item())
    T_top = count_top_rows(hits)
    name, grid, block = engine.last_launch()

    # ---- secondary: the other scaling mode of an N > 1 run, same process, same K / W
    secondary = None
    if distributed and not args.no_secondary and not args.graph and args.scaling == "auto":
        del hd, cols, out
        state["hd"] = state["out"] = None
        hits = None
        torch.cuda.empty_cache()
        hits2, hd2, _, _, out2, _ = build_table("weak")
        state["hd"], state["out"] = hd2, out2
        e2, _ = timed(step)
        tq = torch.tensor([hits2.n_queries], dtype=torch.int64, device=cdev)
        dist.all_reduce(tq)
        secondary = {"scaling": "weak", "value": int(tq.item()) * args.steps / e2 / 1e6, "unit": "Mqueries/s",
                     "ms_per_step": e2 * 1e3 / args.steps, "queries_per_gpu": hits2.n_queries,
                     "note": "every rank holds a full table of its own (per-GPU work fixed as N grows)"}

    # ---- secondary workloads of a default N = 1 run (the headline's numbers are final by now: the table is released first)
    other_workloads = None
    default_run = not (args.queries or args.taxa or args.hits_per_query or args.graph) and args.top_group == "geo" and args.config == "C3" \
        and args.pident == "packed" and not distributed
    if default_run and not args.no_secondary:
        del hd, cols, out
        state["hd"] = state["out"] = None
        hits = None
        torch.cuda.empty_cache()
        other_workloads = run_secondary(args, synth, engine, torch, np, dev, local_rank, custom, {"C3": (tax, eng_tax)})
        e2e = end_to_end_entry()
        if e2e:
            other_workloads.append(e2e)

    ceiling = None
    if rank == 0 and world == 1 and not args.no_secondary:
        if other_workloads is None:                  # (the table is still resident: make room for the probe's buffer)
            state["hd"] = state["out"] = None
            hd = cols = out = hits = None
            torch.cuda.empty_cache()
        ceiling = stream_read_ceiling(torch)
        if ceiling:
            log(f"[bench] stream-read ceiling of this box: {ceiling:.0f} GB/s")

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_q * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        row_bytes = {"f64": 24, "packed64": 28}.get(args.pident, 20)
        alg_bytes = row_bytes * Hn + 8 * (Q + 1) + 32 * Q
        useful = useful_bytes(args.pident, Hn, Q, T_top)
        gbps = lambda b: b / (k_ms * 1e-3) / 1e9
        traffic, traffic_note = None, None
        default_workload = not (args.queries or args.taxa or args.hits_per_query) and args.top_group == "geo" and world == 1
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        tkey = args.config + "-" + args.pident
        if not default_workload:
            traffic_note = "no PMC passes for this workload"
        elif not os.path.exists(tfile):
            traffic_note = "profiles/hbm_traffic.json missing"
        else:
            try:
                ent = json.load(open(tfile)).get(tkey)
                if ent is None:
                    traffic_note = f"no PMC passes for {tkey}"
                elif ent.get("kernel_sha256") != kernel_source_sha256():
                    traffic_note = f"PMC passes of {tkey} were taken with another kernel source (sha256 mismatch): not quoted"
                else:
                    traffic = ent.get("traffic_bytes_per_launch")
            except Exception as e:   # a malformed file must not take the line down
                traffic_note = f"profiles/hbm_traffic.json unreadable: {e}"
        roofline = {"bound": "hbm", "achieved": gbps(useful), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": gbps(useful) / HBM_PEAK_GBPS, "traffic": traffic,
                    "kernel": name, "kernel_ms": k_ms, "kernel_ms_min": float(np.min(kernel_ms)), "kernel_ms_median": float(np.median(kernel_ms)),
                    "launch": {"grid": grid, "block": block},
                    "useful_bytes": useful, "useful_frac": gbps(useful) / HBM_PEAK_GBPS, "top_rows": T_top,
                    "algorithmic_bytes": alg_bytes,
                    "by_formula": {"achieved": gbps(alg_bytes), "frac": gbps(alg_bytes) / HBM_PEAK_GBPS,
                                   "note": f"{row_bytes} B per hit row: every byte of the table, read or not"}}
        if traffic:
            roofline["traffic_frac"] = gbps(traffic) / HBM_PEAK_GBPS
            roofline["traffic_over_useful"] = traffic / useful
        if traffic_note:
            roofline["traffic_note"] = traffic_note
        if ceiling:
            roofline["stream_read_ceiling"] = ceiling          # GB/s, measured in this run on this box (read-only nt stream)
            roofline["frac_of_ceiling"] = gbps(useful) / ceiling
            if traffic:
                roofline["traffic_frac_of_ceiling"] = gbps(traffic) / ceiling
        if pack_info:
            roofline["pack"] = pack_info
        hpq = cfg["hits_per_query"] if cfg["zipf"] is None else "Zipf" + str(cfg["zipf"])
        what = (f"{args.config}: {Q_table} queries x {hpq} hits" +
                (f" per GPU" if scaling == "weak" or world == 1 else f" in ONE table sharded over {world} GPUs (BASELINE config 4)") +
                f", {tax.n}-taxid synthetic taxonomy, strategy {args.strategy}, taxon {args.taxon}, top group {args.top_group}, "
                f"layout {LAYOUT_TEXT[args.pident]}")
        line = {
            "metric": "Mqueries/sec consensus (synthetic outfmt-6 hit table)",
            "value": value, "unit": "Mqueries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE_TEXT[args.pident], "data": "synthetic",
            "config": {"workload": what, "queries_total": total_q, "queries_rank0": Q, "hit_rows_rank0": Hn, "taxids": tax.n,
                       "strategy": args.strategy, "taxon": args.taxon, "top_group": args.top_group, "pident_layout": args.pident,
                       "bytes_per_hit": row_bytes, "seed": hex(seed), "generator_version": synth.GENERATOR_VERSION,
                       "parallelism": f"query-sharded x{world}, taxonomy replicated, no data-path collective",
                       "process_group": backend},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        if secondary:
            line["weak_scaling"] = secondary
        if other_workloads is not None:
            line["secondary"] = other_workloads
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
