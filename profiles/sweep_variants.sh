#!/bin/bash
# Rebuild librt_amd.so with different -D flags (on the GPU box) and time the bench.
# Usage: [BENCH_ARGS="--spp 8"] [ENVS="RT_FUSE=0"] sweep_variants.sh "<flags>" "<flags>" ...
for v in "$@"; do
  make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="$v" 2>&1 | grep -i error
  echo "[$v] ${ENVS:-}: $( ( export ${ENVS:-X=1}; timeout -k 5 ${TMO:-120} python bench.py --no-cpu-baseline --no-count --steps ${STEPS:-4} --warmup 2 $BENCH_ARGS 2>/dev/null ) | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.*}}\|frame_checksum.: .[0-9a-f]*' | cut -c1-200 | tr '\n' ' ')"
done
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc
