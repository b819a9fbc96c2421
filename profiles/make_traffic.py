#!/usr/bin/env python3
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes (profiles/traffic_passes.sh).

Per MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane)
coalesced reads, WRITE_SIZE is exact for 16 B/lane stores.  The correction is calibrated here on a
kernel of this very run whose traffic is known exactly: k_accumulate reads frames*pixels*16 B of
samples + pixels*16 B of accumulator and writes pixels*16 B (all 16 B/lane streaming).
Usage: make_traffic.py <pmc dir> <width> <height> <spp> <workload>"""
import csv, glob, json, os, sys
root, W, H, spp, workload = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
tot = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, cname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "rtd::" not in k or r["Counter_Name"] != cname:
                continue
            k = k.split("rtd::")[1].split("(")[0]
            d = tot.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}})
            d[cname] += float(r["Counter_Value"])
            d["launches"][cname] += 1
acc = tot["k_accumulate"]
n_acc = acc["launches"]["FETCH_SIZE"]
known_read = n_acc * (spp * W * H * 16 + W * H * 16)
known_write = n_acc * (W * H * 16)
fetch_corr = known_read / (acc["FETCH_SIZE"] * 1024)   # counter unit: KiB
write_corr = known_write / (acc["WRITE_SIZE"] * 1024)
out = {"workload": workload, "width": W, "height": H, "spp": spp,
       "calibration": {"kernel": "k_accumulate", "fetch_correction": round(fetch_corr, 4), "write_correction": round(write_corr, 4),
                       "note": "known bytes / (counter KiB * 1024); the guide predicts 2.0 for 16 B/lane reads and 1.0 for writes"},
       "per_kernel": {}}
for k, d in sorted(tot.items()):
    n = max(1, d["launches"]["FETCH_SIZE"])
    # the guide's correction (x2 on reads) is applied; the calibrated factor is reported beside it
    rd = d["FETCH_SIZE"] * 1024 * 2.0 / n
    wr = d["WRITE_SIZE"] * 1024 / max(1, d["launches"]["WRITE_SIZE"])
    out["per_kernel"][k] = {"launches": n, "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr)}
# the timed extend kernels: k_extend<false, *> (first argument: counting; second: the round-0 variant without a queue)
tot_b, tot_n = 0, 0
for k, v in out["per_kernel"].items():
    if k.startswith("k_extend<false"):
        tot_b += (v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]) * v["launches"]
        tot_n += v["launches"]
ext = {"launches": tot_n, "hbm_bytes_per_launch": int(tot_b / max(1, tot_n))}
out["hbm_bytes_per_launch"] = ext["hbm_bytes_per_launch"]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json"), "w"), indent=1)
print(json.dumps(out["calibration"]), out["hbm_bytes_per_launch"], ext)
