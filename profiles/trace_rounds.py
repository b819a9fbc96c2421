#!/usr/bin/env python3
"""Per-launch durations of the LAST bench step in a rocprofv3 kernel trace csv, in launch order: python3 profiles/trace_rounds.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rtd::", "") for r in rows]
# the last step starts at the last generate launch
gen = [i for i, n in enumerate(names) if n.startswith("k_generate")]
start = gen[-1]
t0 = int(rows[start]["Start_Timestamp"])
tot = {}
for r, n in list(zip(rows, names))[start:]:
    if not n.startswith("k_"): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%9.1f us  +%8.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, n))
    tot[n] = tot.get(n, 0) + d
print({k: round(v / 1e3, 3) for k, v in tot.items()})
