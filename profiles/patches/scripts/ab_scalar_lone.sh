#!/bin/bash
# A/B on one box: a lone lane's records through the scalar cache (RT_SCALAR_LONE=1) or through the vector path (0)
for v in "-DRT_SCALAR_LONE=0" "-DRT_SCALAR_LONE=1"; do
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  echo "== $v"
  timeout -k 5 200 python profiles/tick_time.py 2>&1 | grep "never" | cut -c1-200
  for a in "--spp 8" "--spp 64"; do
    echo "bench $a: $(timeout -k 5 200 python bench.py --no-cpu-baseline --no-count --steps 4 --warmup 2 $a 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.*}}\|frame_checksum.: .[0-9a-f]*' | cut -c1-200 | tr '\n' ' ')"
  done
done
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc
