#!/bin/bash
# Where do a traversal wave's cycles go?  Rebuilds librt_amd.so with -DRT_SECTION_PROBE (ON THE GPU BOX), renders one
# step of the bench workload through the plain round loop and prints, per traversal launch, the share of wave cycles per
# section of trace_persistent (rt_scene_dev.h); then restores the product build.
#   bash profiles/section_probe.sh [bench args]  >  profiles/<tag>_section_probe.txt
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="-DRT_SECTION_PROBE" 2>&1 | grep -i error
RT_FUSE=0 python bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" 2>&1 | grep "section probe"
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc 2>&1 | grep -i error
exit 0
