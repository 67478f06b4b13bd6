#!/usr/bin/env python3
"""Copies the judged summaries of gpurun_out/profiles_<tag>/ into profiles/ (tracked) and derives the HBM
traffic figure bench.py quotes: traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes), the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE = TCC_EA0_RDREQ x 64 B while every read request of this kernel
is a 128-byte request: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ)."""
import json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
config = sys.argv[2] if len(sys.argv) > 2 else "C3-packed"   # key = config, "-milli" suffix for the 20 B/hit layout
src = f"gpurun_out/profiles_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/kernel_stats_blu.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/pmc/pmc_summary.json", f"profiles/{tag}_pmc_summary.json")
shutil.copy(f"{src}/bench_under_trace.json", f"profiles/{tag}_bench_under_kernel_trace.json")
if os.path.exists(f"{src}/stream_read_ceiling.txt"):
    shutil.copy(f"{src}/stream_read_ceiling.txt", f"profiles/{tag}_stream_read_ceiling.txt")
pmc = json.load(open(f"{src}/pmc/pmc_summary.json"))
k = next(v for n, v in pmc.items() if "stream_kernel" in n)
fetch, write = k["FETCH_SIZE"] * 1024, k["WRITE_SIZE"] * 1024
traffic = 2 * fetch + write
path = "profiles/hbm_traffic.json"
d = json.load(open(path)) if os.path.exists(path) else {}
d[config] = {"round": tag, "kernel": "blu_consensus_stream_kernel", "traffic_bytes_per_launch": traffic,
             "FETCH_SIZE_bytes_uncorrected": fetch, "WRITE_SIZE_bytes": write,
             "TCC_EA0_RDREQ": k.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_128B": k.get("TCC_EA0_RDREQ_128B_sum"),
             "TCC_EA0_WRREQ_64B": k.get("TCC_EA0_WRREQ_64B_sum"),
             "note": "read side = TCC_EA0_RDREQ x 128 B; Infinity-Cache hits are counted (the guide: fabric-side counters)"}
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d[config], indent=1))
