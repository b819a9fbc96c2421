#!/bin/bash
# Steps per nearest-hit ray of every extend launch of one bench step (plain round loop), by the material type of the
# surface the ray left (0 diffuse, 2 metal, 3 glass ...; 7 = camera ray): which rays are the long ones that a traversal
# launch waits for at its end?  Rebuilds with -DRT_STEP_COUNT ON THE GPU BOX, restores the product build afterwards.
#   bash profiles/step_histogram.sh [bench args] > profiles/<tag>_step_histogram.txt
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA="-DRT_STEP_COUNT" 2>&1 | grep -i " error"
RT_FUSE=0 python bench.py --no-cpu-baseline --steps 1 --warmup 0 "$@" 2>&1 | grep "step count\|instance entries\|left a surface" | tail -80
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc 2>&1 | grep -i " error"
exit 0
