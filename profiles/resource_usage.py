#!/usr/bin/env python3
"""Registers / occupancy / scratch / LDS of every kernel: `make -C ray-and-pathtracer_amd/csrc resource-usage 2>&1 | python3 profiles/resource_usage.py [filter]`"""
import re, subprocess, sys
txt = sys.stdin.read()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
names = list(rows)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for n, d in zip(names, dem):
    short = re.sub(r"\(.*", "", d).replace("void ", "").replace("rtd::", "")
    if pat and not re.search(pat, short):
        continue
    r = rows[n]
    print("%-42s vgpr %3d agpr %3d sgpr %3d waves %d scratch %4d lds %6d" % (short[:42], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("Occupancy", -1), r.get("ScratchSize", -1), r.get("LDS", -1)))
