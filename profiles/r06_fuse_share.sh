#!/bin/bash
# The 1/8 share (16.6 M samples) with one traversal launch per round (RT_FUSE=1: k_traverse_s) against the default (two streams) and one kernel at a time
mkdir -p gpurun_out/r06
for rep in 1 2; do for f in -1 1 0; do RT_FUSE=$f timeout -k 10 120 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-count --emulate-world 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('RT_FUSE=$f  %8.3f ms  %s  %s' % (d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))"; done; done 2>&1 | tee gpurun_out/r06/fuse_share.txt
