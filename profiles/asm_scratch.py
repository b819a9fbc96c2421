#!/usr/bin/env python3
"""Where does a kernel spill?  python3 profiles/asm_scratch.py /tmp/rt_api.s <mangled-name-substring>
Prints, per basic block of the kernel that holds scratch accesses, the number of scratch loads / stores, the block's
instruction count and the vector-memory loads in it (a pair step is four global_load_dwordx4), with the loop headers
(.LBB labels that are branch targets from below) marked -- enough to see whether spills sit in the hot loop or around rare code.
(Assembly from: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S rt_api.hip)"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
blocks, cur = [], ["entry", start, []]
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur); cur = [m.group(1), i, []]
    elif l.startswith("\t") and not l.strip().startswith((";", ".")):
        cur[2].append(l.strip())
blocks.append(cur)
index = {b[0]: k for k, b in enumerate(blocks)}
back = set()
for k, b in enumerate(blocks):
    for ins in b[2]:
        m = re.search(r"(s_cbranch\w*|s_branch)\s+(\.LBB\d+_\d+)", ins)
        if m and index.get(m.group(2), 1 << 30) <= k: back.add(m.group(2))
tot_l = tot_s = 0
for k, b in enumerate(blocks):
    ld = sum(1 for x in b[2] if x.startswith("scratch_load"))
    stc = sum(1 for x in b[2] if x.startswith("scratch_store"))
    gl = sum(1 for x in b[2] if x.startswith("global_load_dwordx4"))
    ds = sum(1 for x in b[2] if x.startswith("ds_"))
    tot_l += ld; tot_s += stc
    if ld or stc or b[0] in back:
        print("%-12s line %7d  insts %4d  scratch ld %2d st %2d  global x4 %2d  ds %2d %s" % (b[0], b[1], len(b[2]), ld, stc, gl, ds, "<- loop header" if b[0] in back else ""))
print("total scratch loads %d stores %d, blocks %d, instructions %d" % (tot_l, tot_s, len(blocks), sum(len(b[2]) for b in blocks)))
