#!/bin/bash
# Same-box A/B of bench.py over trees built before the call (the repository itself, copies under _v/):
#   [STEPS=8 WARMUP=2 REPS=2] bash profiles/bisect.sh "<dir>[@ENV=VAL[,ENV=VAL]] ..." [bench args]        alternating repetitions
cd "$(dirname "$0")/.."
DIRS=$1; shift
for rep in $(seq 1 ${REPS:-2}); do for spec in $DIRS; do
  d=${spec%%@*}; envs=""; [ "$spec" != "$d" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  ( cd $d && env $envs timeout -k 10 ${TMO:-600} python3 bench.py --steps ${STEPS:-8} --warmup ${WARMUP:-2} --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-34s %9.3f ms  %s  %s' % ('$spec', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" )
done; done
