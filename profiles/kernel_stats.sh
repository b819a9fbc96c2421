#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench for a set of environments, ONE box: bash profiles/kernel_stats.sh <tag> "<ENV=.. ENV=..>" ["<env 2>" ...]
# writes gpurun_out/<tag>_<i>_kernel_stats.csv (+ the per-launch trace) and prints the per-kernel table (ms per step)
set -u
TAG=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
I=0
STEPS=${STEPS:-3}
ARGS=${BENCH_ARGS:-}
for ENVS in "$@"; do
  OUT=gpurun_out/${TAG}_${I}
  rm -rf $OUT
  ( export $ENVS; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps $STEPS --warmup 1 --no-cpu-baseline $ARGS > $OUT.log 2>&1 )
  cp $OUT/*/*kernel_stats.csv gpurun_out/${TAG}_${I}_kernel_stats.csv 2>/dev/null
  cp $OUT/*/*kernel_trace.csv gpurun_out/${TAG}_${I}_kernel_trace.csv 2>/dev/null
  echo "== $ENVS"
  python3 - <<PY
import csv, json, re
steps = $STEPS + 2   # counting pass + warmup + timed
rows = list(csv.DictReader(open("gpurun_out/${TAG}_${I}_kernel_stats.csv")))
tot = 0
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("rtd::", "")
    if not name.startswith("k_"): continue
    ms = float(r["TotalDurationNs"]) / 1e6
    print("  %-34s calls %4d  total %9.3f ms  avg %8.1f us" % (name, int(r["Calls"]), ms, float(r["AverageNs"]) / 1e3))
try:
    line = [l for l in open("$OUT.log") if l.startswith("{")][-1]
    d = json.loads(line)
    print("  ms_per_step %.3f  checksum %s  kernel_ms %s" % (d["ms_per_step"], d["frame_checksum"], d["roofline"]["kernel_ms_per_step"]))
except Exception as e:
    print("  no bench line:", e)
PY
  rm -rf $OUT
  I=$((I+1))
done
