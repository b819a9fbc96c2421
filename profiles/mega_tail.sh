#!/bin/bash
# Where a Whitted Tick's single launch spends its time: queue dry vs last wave out (-DRT_TAIL_PROBE build, on the GPU box).
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc EXTRA=-DRT_TAIL_PROBE 2>&1 | grep -i error
timeout -k 5 200 python profiles/tick_time.py 2>&1
make -s -C ray-and-pathtracer_amd/csrc clean; make -s -C ray-and-pathtracer_amd/csrc
