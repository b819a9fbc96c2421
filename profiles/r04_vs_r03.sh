#!/bin/bash
# NOTE (round 5): this script sweeps RT_REFILL / RT_STEPMIN / RT_PAIRAGAIN / RT_DRAIN_LANES through the ENVIRONMENT, as the library read them when it was
# written.  They are compile-time constants since round 4 (csrc/rt_scene_dev.h; the library prints a warning when it sees one set): re-running
# it as it is gives identical rows.  A sweep is a rebuild per value now: profiles/bisect.sh over trees built with make EXTRA=-DRT_...=N.
# Same-box A/B of the tree as it is against round 3's final tree (a git worktree at _r03/, built before the call):
#   bash profiles/r04_vs_r03.sh -> gpurun_out/r04_vs_r03.txt
cd "$(dirname "$0")/.."
OUT=$PWD/gpurun_out/r04_vs_r03.txt
: > $OUT
run() { # dir label args...
  local dir=$1 label=$2; shift 2
  ( cd $dir && timeout -k 10 300 python3 bench.py --warmup 2 --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-44s %8.3f ms  %s  %s' % ('$label', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" ) >> $OUT 2>&1
}
for rep in 1 2; do
  echo "== full frame, rep $rep" >> $OUT
  run _r03 "r03" --steps 8
  run . "r04" --steps 8
  echo "== 1/8 share (--emulate-world 8), rep $rep" >> $OUT
  run _r03 "r03 (default: 'gate' that never waited)" --steps 12 --emulate-world 8
  RT_FUSE=2 run . "r04 RT_FUSE=2" --steps 12 --emulate-world 8
  RT_FUSE=3 run . "r04 RT_FUSE=3" --steps 12 --emulate-world 8
  RT_FUSE=2 RT_DRAIN_LANES=64 run . "r04 RT_FUSE=2 RT_DRAIN_LANES=64" --steps 12 --emulate-world 8
  RT_FUSE=2 RT_DRAIN_LANES=0 run . "r04 RT_FUSE=2 RT_DRAIN_LANES=0" --steps 12 --emulate-world 8
  echo "== --spp 16, rep $rep" >> $OUT
  run _r03 "r03" --steps 10 --spp 16
  RT_FUSE=2 run . "r04 RT_FUSE=2" --steps 10 --spp 16
done
cat $OUT
