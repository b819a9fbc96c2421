#!/bin/bash
# Sweep the traversal scheduling thresholds (and the 4-wide occlusion walk) on the bench workload, all on one box.
# Usage: profiles/sweep_tuning.sh > log
run() { echo "$1: $(env $1 python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.: {[^}]*}' | tr '\n' ' ')"; }
run "RT_NONE=0"
for rf in 8 12 24; do run "RT_REFILL=$rf"; done
for ra in 16 24 48; do run "RT_REFILL_ANY=$ra"; done
for sm in 6 8 16 20; do run "RT_STEPMIN=$sm"; done
for pa in 8 12 24; do run "RT_PAIRAGAIN=$pa"; done
run "RT_WIDE=1"
run "RT_NONE=0"
