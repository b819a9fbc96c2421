#!/bin/bash
# Sweep the traversal scheduling thresholds on the bench workload.  Usage: profiles/sweep_tuning.sh > log
for sm in 8 12 16 24 32; do for rf in 16 24 32; do
 echo "stepmin $sm refill $rf: $(RT_STEPMIN=$sm RT_REFILL=$rf python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | grep -o 'kernel_ms_per_step.*}}')"
done; done
