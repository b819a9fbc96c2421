#!/bin/bash
# k_accumulate as a stream with eight loads in flight and a branch-free gamma, against round 5's serial loop (_v/oldacc): parity subset, then the
# kernel's own time from rocprofv3 --kernel-trace --stats on the default workload and on config 5's batch shape (3840x2160 x 32 spp).
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
for d in ${DIRS:-. _v/oldacc}; do for args in "--steps 3" "--workload config5 --spp 32 --steps 2"; do
  rm -rf gpurun_out/r06/acc_stats
  ( cd $d && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r06/acc_stats -- python3 bench.py --warmup 1 --no-cpu-baseline --no-count $args > /root/repo/gpurun_out/r06/acc_run.log 2>&1 )
  echo "== $d $args: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r06/acc_run.log | tail -1) $(grep -o '"frame_checksum": "[0-9a-f]*"' gpurun_out/r06/acc_run.log | tail -1)"
  python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/r06/acc_stats/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_accumulate" in r["Name"] or "k_shade_s" in r["Name"] or "k_light_s" in r["Name"]:
            print("   %-60s calls %3s  avg %10.1f us  total %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done; done 2>&1 | tee gpurun_out/r06/acc_ab.txt
