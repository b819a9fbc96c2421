#!/bin/bash
# VMEM instructions / texture-addresser busy of the traversal kernels with the 32-byte relative pair records on and off (the six-wave
# build they fit in, _v/p32w6, built before the call): bash profiles/r04_pair32_counters.sh >> gpurun_out/r04_ab_pair32.txt
set -u
export TMPDIR=/tmp
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for on in 0 1; do
  OUT=$ROOT/gpurun_out/p32c_$on
  rm -rf $OUT; mkdir -p $OUT
  ( cd $ROOT/_v/p32w6 && export RT_PAIR32=$on && for pass in "ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "sq3 SQ_INSTS_VMEM SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
      set -- $pass; name=$1; shift
      timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-count > $OUT/$name.log 2>&1 || echo "pass $name failed"
    done )
  python3 $ROOT/profiles/pmc_summary.py $OUT > $OUT/summary.json
  python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
print("== RT_PAIR32=$on (six-wave build with the records compiled in), one step of the full frame")
for k, v in sorted(d.items()):
    if not (k.startswith("k_extend_s") or k.startswith("k_connect_s")): continue
    c = v["counters"]; cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    print("  %-44s %7.2f ms  VMEM insts %6.1f M  VALU insts %7.1f M  lane-ops %6.1f G  TA busy avg %.3f max %.3f" % (k, v["ms_by_pass"].get("sq3", 0), c.get("SQ_INSTS_VMEM", 0) / 1e6, c.get("SQ_INSTS_VALU", 0) / 1e6, c.get("SQ_THREAD_CYCLES_VALU", 0) / 1e9, c.get("TA_BUSY_avr", 0) / max(cyc, 1), c.get("TA_BUSY_max", 0) / max(cyc, 1)))
PY
  rm -rf $OUT/ta1 $OUT/sq3
done
