#!/bin/bash
# NOTE (round 5): this script sweeps RT_REFILL / RT_STEPMIN / RT_PAIRAGAIN / RT_DRAIN_LANES through the ENVIRONMENT, as the library read them when it was
# written.  They are compile-time constants since round 4 (csrc/rt_scene_dev.h; the library prints a warning when it sees one set): re-running
# it as it is gives identical rows.  A sweep is a rebuild per value now: profiles/bisect.sh over trees built with make EXTRA=-DRT_...=N.
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
