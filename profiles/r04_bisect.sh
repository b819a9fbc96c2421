#!/bin/bash
# Same-box bisect of the full-frame step over worktrees built before the call: bash profiles/r04_bisect.sh "<dir> <dir> ..." [bench args]
cd "$(dirname "$0")/.."
DIRS=$1; shift
for rep in 1 2; do for d in $DIRS; do
  ( cd $d && timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-24s %8.3f ms  %s  %s' % ('$d', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" )
done; done
