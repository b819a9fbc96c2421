#!/bin/bash
# VGPRs / scratch / occupancy of the kernels of librt_amd.so as the compiler reports them (no GPU needed):
#   bash profiles/resource_usage.sh [EXTRA="-D..."] [pattern]
cd "$(dirname "$0")/../ray-and-pathtracer_amd/csrc"
PAT=${2:-k_}
make -s resource-usage EXTRA="${1:-}" 2>&1 | python3 -c "
import re, sys, subprocess
txt = sys.stdin.read()
cur = None; rows = {}
for line in txt.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur).replace('void ', '').replace('rtd::', '')
        rows[cur] = {}
        continue
    for key in ('VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs', 'LDS Size [bytes/block]'):
        m = re.search(re.escape(key) + r': (\d+)', line)
        if m and cur: rows[cur][key.split(' ')[0]] = int(m.group(1))
for k, v in sorted(rows.items()):
    if '$PAT' in k: print('%-60s VGPR %3d  scratch %4d  waves %d  LDS %5d' % (k, v.get('VGPRs', -1), v.get('ScratchSize', -1), v.get('Occupancy', -1), v.get('LDS', -1)))
"
