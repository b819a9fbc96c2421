#!/bin/bash
# The sampler's reward atomic with and without its return value (the overflow check of ADVICE r5): config 5 with the sampler on, one box.
mkdir -p gpurun_out/r06
for rep in 1 2; do for d in . _v/qnoret; do ( cd $d && timeout -k 10 300 python3 bench.py --workload config5 --qlearn 32 --steps 2 --warmup 1 --no-cpu-baseline --no-count 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-12s %9.3f ms  %s  %s' % ('$d', d['ms_per_step'], d['frame_checksum'], d['roofline']['kernel_ms_per_step']))" ); done; done 2>&1 | tee gpurun_out/r06/qlearn_ab.txt
