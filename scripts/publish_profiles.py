#!/usr/bin/env python3
"""usage: scripts/publish_profiles.py <tag> <key>      e.g.  r02 C3-packed | r02c5 C5-packed | r02f64 C3-f64

Copies the judged summaries of gpurun_out/profiles_<tag>/ into profiles/ (tracked) and derives the HBM traffic figure
bench.py quotes under <key> (= "<config>-<layout>"): traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes) summed over the
blu_consensus_* kernels of one run — the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE =
TCC_EA0_RDREQ x 64 B while the read requests of these kernels are 128-byte requests: the summary keeps TCC_EA0_RDREQ and
TCC_EA0_RDREQ_128B side by side so that the premise can be checked).  The entry carries the sha256 of the kernel source
the counters were taken with; bench.py quotes the figure only while that source is unchanged."""
import json, os, shutil, sys
tag, key = sys.argv[1], sys.argv[2]
src = f"gpurun_out/profiles_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/kernel_stats_blu.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"{src}/pmc/pmc_summary.json", f"profiles/{tag}_pmc_summary.json")
shutil.copy(f"{src}/bench_under_trace.json", f"profiles/{tag}_bench_under_kernel_trace.json")
if os.path.exists(f"{src}/stream_read_ceiling.txt"):
    shutil.copy(f"{src}/stream_read_ceiling.txt", f"profiles/{tag}_stream_read_ceiling.txt")
sha = open(f"{src}/kernel_sha256.txt").read().strip()
pmc = json.load(open(f"{src}/pmc/pmc_summary.json"))
kernels = {}
traffic = 0.0
for name, k in pmc.items():
    if "blu_consensus" not in name and "blu_classify" not in name:
        continue
    fetch, write = k["FETCH_SIZE"] * 1024, k["WRITE_SIZE"] * 1024
    # (the stream kernel exists with and without the bit-score ring; the kind a table does not use shows up once, returning at once)
    short = ("stream" if ", true>" in name else "stream_noring") if "stream_kernel" in name else ("long" if "long_kernel" in name else name)
    kernels[short] = {"FETCH_SIZE_bytes_uncorrected": fetch, "WRITE_SIZE_bytes": write, "traffic_bytes": 2 * fetch + write,
                      "TCC_EA0_RDREQ": k.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_RDREQ_128B": k.get("TCC_EA0_RDREQ_128B_sum"),
                      "TCC_EA0_RDREQ_64B": k.get("TCC_EA0_RDREQ_64B_sum"), "TCC_EA0_RDREQ_32B": k.get("TCC_EA0_RDREQ_32B_sum"),
                      "TCC_EA0_WRREQ_64B": k.get("TCC_EA0_WRREQ_64B_sum"), "TCC_HIT": k.get("TCC_HIT_sum"), "TCC_MISS": k.get("TCC_MISS_sum")}
    if "blu_consensus" in name:   # (blu_classify_tasks runs on the first call on a handle and every 64th: listed, not summed)
        traffic += 2 * fetch + write
path = "profiles/hbm_traffic.json"
d = json.load(open(path)) if os.path.exists(path) else {}
d[key] = {"round": tag, "kernel_sha256": sha, "bench_args": open(f"{src}/bench_args.txt").read().strip(),
          "traffic_bytes_per_launch": traffic, "kernels": kernels,
          "note": "per launch of blu_consensus_run (both kernels); read side = 2 x FETCH_SIZE (128-byte requests tallied at 64 B); "
                  "Infinity-Cache hits are counted (the guide: fabric-side counters)"}
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d[key], indent=1))
