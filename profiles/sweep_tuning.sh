#!/bin/bash
# Sweep the traversal scheduling thresholds on the bench workload.  Usage: profiles/sweep_tuning.sh > log
run() { echo "$1: $(env $1 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | grep -o 'kernel_ms_per_step.*}}')"; }
for rf in 8 12 16 24; do run "RT_REFILL=$rf"; done
for ra in 12 16 24 32 48; do run "RT_REFILL_ANY=$ra"; done
for sm in 6 8 12 16 20; do run "RT_STEPMIN=$sm"; done
for pa in 8 16 24; do run "RT_PAIRAGAIN=$pa"; done
