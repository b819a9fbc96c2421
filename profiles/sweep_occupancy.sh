#!/bin/bash
# Rebuild with different launch bounds for the shading kernels (on the GPU box) and time the bench.
for v in "4 2 4" "5 3 5" "6 4 6" "8 4 8" "4 4 4" "6 3 4"; do
  set -- $v
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="-DRT_SHADE_WAVES=$1 -DRT_LIGHT_WAVES=$2 -DRT_FINISH_WAVES=$3" 2>&1 | grep -i error
  echo "shade $1 light $2 finish $3: $(python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.*}}' | tr '\n' ' ')"
done
