#!/bin/bash
# A/B of one environment knob on the same GPU box: ab_env.sh VAR "v1 v2 ..." [bench args]; alternates the values twice.
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for v in $VALS; do
  echo "$VAR=$v: $(env $VAR=$v python bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.: {[^}]*}\|frame_checksum.: .[0-9a-f]*' | tr '\n' ' ')"
done; done
