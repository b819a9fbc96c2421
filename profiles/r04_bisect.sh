#!/bin/bash
# Same-box A/B of bench.py over trees built before the call (the repository itself, git worktrees / copies under _v/, _r03/):
#   bash profiles/r04_bisect.sh "<dir>[@ENV=VAL[,ENV=VAL]] ..." [bench args]        two alternating repetitions
cd "$(dirname "$0")/.."
DIRS=$1; shift
for rep in 1 2; do for spec in $DIRS; do
  d=${spec%%@*}; envs=""; [ "$spec" != "$d" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  ( cd $d && env $envs timeout -k 10 300 python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %8.3f ms  %s  %s' % ('$spec', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" )
done; done
