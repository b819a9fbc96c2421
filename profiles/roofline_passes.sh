#!/bin/bash
# Everything bench.py's roofline block needs from counters, for the kernels as they are in the tree now:
#   PMC passes of the full default workload (separate rocprofv3 --pmc runs, counters only) -> summary -> roofline_pmc.json
# and the kernel-trace statistics of the same command.  Run ON THE GPU BOX from the repository root:
#   bash profiles/roofline_passes.sh <tag> [worlds]        e.g.  r03_final "1 2 4 8"
# For every number of ranks N in 'worlds' the passes run bench.py --emulate-world N (rank 0's rows of the N-rank shard, one GPU).
# Writes gpurun_out/<tag>_w<N>_pmc/ (raw), gpurun_out/<tag>_w<N>_pmc_summary.json, gpurun_out/roofline_pmc.json, gpurun_out/<tag>_kernel_stats.csv;
# copy the summaries and roofline_pmc.json into profiles/ afterwards.
set -u
TAG=$1
WORLDS=${2:-1}
export TMPDIR=/tmp
# the committed profiles/roofline_pmc.json is replaced only after every world's passes succeeded
export RT_ROOFLINE_PMC_OUT=gpurun_out/roofline_pmc.json.tmp
rm -f $RT_ROOFLINE_PMC_OUT
for N in $WORLDS; do
  OUT=gpurun_out/${TAG}_w${N}_pmc
  EMU=""; [ "$N" != "1" ] && EMU="--emulate-world $N"
  bash profiles/pmc_passes.sh $OUT --steps 2 --warmup 0 --no-cpu-baseline --no-count $EMU
  python3 profiles/pmc_summary.py $OUT > gpurun_out/${TAG}_w${N}_pmc_summary.json
  KH=$(grep -o '"kernel_hash": "[0-9a-f]*"' $OUT/sq1.log | head -1 | grep -o '[0-9a-f]\{16\}')
  if [ -z "$KH" ]; then echo "no kernel_hash in $OUT/sq1.log (the bench did not finish): aborting, profiles/roofline_pmc.json untouched"; exit 1; fi
  python3 profiles/valu_roofline.py gpurun_out/${TAG}_w${N}_pmc_summary.json config3 1920 1080 64 $KH $N > gpurun_out/${TAG}_w${N}_roofline.txt || { echo "valu_roofline.py failed for world $N: aborting"; exit 1; }
  find $OUT -name "*.csv" -delete   # the raw per-dispatch tables are large: the summaries are what is kept
done
mv $RT_ROOFLINE_PMC_OUT profiles/roofline_pmc.json
cp profiles/roofline_pmc.json gpurun_out/roofline_pmc.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_stats.log 2>&1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_stats
grep -o '"frac_of_peak_lane_ops": [0-9.]*' gpurun_out/roofline_pmc.json | head -12
