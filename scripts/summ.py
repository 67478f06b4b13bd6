#!/usr/bin/env python3
"""usage: scripts/summ.py <tag> name...  — one line per gpurun_out/<tag>/<name>.out bench JSON."""
import json, sys
tag = sys.argv[1]
for n in sys.argv[2:]:
    try:
        d = json.loads(open(f"gpurun_out/{tag}/{n}.out").read().strip().splitlines()[-1]); r = d["roofline"]
        print("%-8s value %8.0f Mq/s  kernel_ms %.3f  useful_frac %.3f  by_formula %.3f" % (n, d["value"], r["kernel_ms"], r["frac"], r["by_formula"]["frac"]))
    except Exception as e:
        print(n, "ERR", e, open(f"gpurun_out/{tag}/{n}.err").read()[-400:])
