#!/bin/bash
# Tick / Whitted frame latency for compile-time variants on one box: tick_ab.sh "<flags>" "<flags>" ...
for v in "$@"; do
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="$v" 2>&1 | grep -i " error"
  echo "== [$v]"; python profiles/tick_time.py 2>&1 | grep "never"; python bench.py --no-cpu-baseline --spp 8 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*'
done
