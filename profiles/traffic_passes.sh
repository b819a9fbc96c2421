#!/bin/bash
# HBM traffic of the bench workload: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots),
# counters only.  Usage: profiles/traffic_passes.sh <outdir>
set -u
OUT=$1
export TMPDIR=/tmp
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/$c.log 2>&1 || echo "pass $c failed"
done
