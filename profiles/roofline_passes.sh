#!/bin/bash
# Everything bench.py's roofline block needs from counters, for the kernels as they are in the tree now:
#   PMC passes of the full default workload (separate rocprofv3 --pmc runs, counters only) -> summary -> roofline_pmc.json
# and the kernel-trace statistics of the same command.  Run ON THE GPU BOX from the repository root:
#   bash profiles/roofline_passes.sh <tag>        e.g.  r02_v13
# Writes gpurun_out/<tag>_pmc/ (raw), profiles/<tag>_pmc_summary.json, profiles/roofline_pmc.json, profiles/<tag>_kernel_stats.csv.
set -u
TAG=$1
export TMPDIR=/tmp
OUT=gpurun_out/${TAG}_pmc
bash profiles/pmc_passes.sh $OUT --steps 2 --warmup 0 --no-cpu-baseline
python3 profiles/pmc_summary.py $OUT > profiles/${TAG}_pmc_summary.json
KH=$(grep -o '"kernel_hash": "[0-9a-f]*"' $OUT/sq1.log | head -1 | grep -o '[0-9a-f]\{16\}')
python3 profiles/valu_roofline.py profiles/${TAG}_pmc_summary.json config3 1920 1080 64 $KH > gpurun_out/${TAG}_roofline_pmc.txt
cp profiles/${TAG}_pmc_summary.json profiles/roofline_pmc.json gpurun_out/
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
tail -3 gpurun_out/${TAG}_roofline_pmc.txt
