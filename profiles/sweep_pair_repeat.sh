#!/bin/bash
# Rebuild librt_amd.so with different RT_PAIR_REPEAT values (on the GPU box) and time the bench.
for n in 2 3 4 5 6; do
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA=-DRT_PAIR_REPEAT=$n 2>&1 | grep -i error
  echo "RT_PAIR_REPEAT $n: $(RT_PAIRAGAIN=16 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.*}}' | tr '\n' ' ')"
done
