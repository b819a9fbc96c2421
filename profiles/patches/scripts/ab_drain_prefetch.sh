#!/bin/bash
# A/B on one box: the drain's line touches (RT_DRAIN_PREFETCH, rt_scene_dev.h) off / on
for v in "-DRT_DRAIN_PREFETCH=0" "-DRT_DRAIN_PREFETCH=1"; do
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  echo "== $v"
  timeout -k 5 200 python profiles/lone_step.py 2>&1 | tail -3
  timeout -k 5 200 python profiles/tick_time.py 2>&1 | grep "never" | cut -c1-200
  for a in "--spp 8" "--spp 64"; do
    echo "bench $a: $(timeout -k 5 200 python bench.py --no-cpu-baseline --no-count --steps 4 --warmup 2 $a 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.*}}\|frame_checksum.: .[0-9a-f]*' | cut -c1-140 | tr '\n' ' ')"
  done
done
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc
