#!/bin/bash
# Build the kernels with index checks on the shared structures (csrc/rt_scene_dev.h RT_CHECK) and run the GPU test
# tier once; restores the product build afterwards.  Usage (on the GPU box): bash profiles/debug_checks.sh <logfile>
LOG=${1:-gpurun_out/debug_checks.log}
make -s -C ray-and-pathtracer_amd/csrc clean
make -s -C ray-and-pathtracer_amd/csrc EXTRA=-DRT_DEBUG_CHECKS 2>&1 | grep -i "error"
python -m pytest tests -m gpu -x -q -k "not fuzz" > $LOG 2>&1
tail -3 $LOG
make -s -C ray-and-pathtracer_amd/csrc clean
make -s -C ray-and-pathtracer_amd/csrc 2>&1 | grep -i "error"
exit 0
