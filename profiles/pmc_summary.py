#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (profiles/pmc_passes.sh) per kernel: sum of each counter over the
launches of the non-counting kernel variants, plus launch count and total duration."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    pas = f.split(os.sep)[-3]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rtd::" not in k or ("<true" in k and "k_shade_s" not in k):  # "<true": the counting variants -- except k_shade_s<QL>, the Q-learning sampler's kernel
            continue
        k = k.split("rtd::")[1].split("(")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (pas, r["Dispatch_Id"])
        if key not in seen[k]:
            seen[k].add(key)
            dur[k][pas] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k in sorted(tot):
    n = {p: len([1 for (pp, _) in seen[k] if pp == p]) for p in dur[k]}
    out[k] = {"launches": max(n.values()), "ms_by_pass": {p: round(v / 1e6, 3) for p, v in dur[k].items()}, "counters": dict(tot[k])}
json.dump(out, sys.stdout, indent=1)
