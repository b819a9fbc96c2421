#!/bin/bash
# Per-launch timeline of ONE bench step (the last one) from a rocprofv3 kernel trace: bash profiles/step_timeline.sh <tag> "<ENV=..>" [bench args]
# -> gpurun_out/r05/timeline_<tag>.txt  (kernel, start relative to the step's first launch, duration; us)
TAG=$1; ENVS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
D=$GRAFT_REPO_ROOT/gpurun_out/r05/tl_$TAG
rm -rf $D; mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r05
( cd $GRAFT_REPO_ROOT && export $ENVS && rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --steps 2 --warmup 1 --no-count --no-cpu-baseline "$@" > $D.json 2> /dev/null )
f=$(find $D -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $GRAFT_REPO_ROOT/gpurun_out/r05/timeline_$TAG.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("rtd::", "").replace("void ", "")
gen = [i for i, r in enumerate(rows) if name(r).startswith("k_generate_s")]
first = gen[-1]
last = max(i for i, r in enumerate(rows) if name(r).startswith("k_accumulate"))
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:last + 1]:
    print("%-28s start %9.1f  dur %8.1f" % (name(r), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("step: %.1f us" % ((int(rows[last]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf $D
cat $GRAFT_REPO_ROOT/gpurun_out/r05/timeline_$TAG.txt
