#!/bin/bash
# Per-launch times of a Whitted Tick by tree levels (RT_MEGA_LEVELS=1): rocprofv3 kernel trace of a few Ticks, the launches of the last Tick in order.
cd /tmp && export TMPDIR=/tmp
for scene in mixed_small pretty_tlas; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/lv_$scene
  ( cd $GRAFT_REPO_ROOT && RT_MEGA_LEVELS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lv_$scene -- python3 profiles/whitted_ticks.py $scene 8 > /dev/null 2>&1 )
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/lv_$scene -name "*kernel_trace.csv" | head -1)
  echo "== $scene"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [r for r in rows if "k_whitted_level" in r["Kernel_Name"] or "k_whitted_reduce" in r["Kernel_Name"]]
last = k[-5:] if len(k) >= 5 else k
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print("%-18s start %7.1f us  duration %7.1f us" % (r["Kernel_Name"].split("(")[0].replace("rtd::", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("frame: %.1f us from the first launch's start to the reduce's end" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3))
PY
done
