#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel (one row per dispatch and counter)."""
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["_dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(res, indent=1, sort_keys=True))
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
