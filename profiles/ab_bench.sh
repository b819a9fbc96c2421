#!/bin/bash
# A/B of bench.py under different environments on ONE box: bash profiles/ab_bench.sh "<ENV=.. ENV=..>" ["<env 2>" ...]
# prints ms_per_step, checksum and the per-kind kernel times (HIP events inside the timed steps) per variant
set -u
cd "$(dirname "$0")/.."
STEPS=${STEPS:-5}
for ENVS in "$@"; do
  ( export $ENVS; timeout -k 10 300 python3 bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-count ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-60s %8.3f ms  %s  %s' % ('$ENVS', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" )
done
