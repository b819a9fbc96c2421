#!/bin/bash
# NOTE (round 5): this script sweeps RT_REFILL / RT_STEPMIN / RT_PAIRAGAIN / RT_DRAIN_LANES through the ENVIRONMENT, as the library read them when it was
# written.  They are compile-time constants since round 4 (csrc/rt_scene_dev.h; the library prints a warning when it sees one set): re-running
# it as it is gives identical rows.  A sweep is a rebuild per value now: profiles/bisect.sh over trees built with make EXTRA=-DRT_...=N.
# Round 4, fourth batch on one box: the scheduling thresholds re-swept on the 64-register / eight-wave kernels, and the backend's
# max-memory-clause scheduling strategy.  bash profiles/r04_batch4.sh -> gpurun_out/r04_batch4.txt
cd "$(dirname "$0")/.."
run() { echo "$1: $(env $1 python3 bench.py --no-cpu-baseline --no-count --steps 6 --warmup 2 $2 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.: {[^}]*}' | tr '\n' ' ')"; }
{
echo "== thresholds, full frame"
run "RT_NONE=0"
for rf in 8 12 24 32; do run "RT_REFILL=$rf"; done
for ra in 16 32 48; do run "RT_REFILL_ANY=$ra"; done
for sm in 4 6 12 16; do run "RT_STEPMIN=$sm"; done
for sm in 4 12 16; do run "RT_STEPMIN_ANY=$sm"; done
for pa in 8 12 24 32; do run "RT_PAIRAGAIN=$pa"; done
for pa in 8 24 32; do run "RT_PAIRAGAIN_ANY=$pa"; done
run "RT_NONE=0"
echo "== thresholds, 1/8 share"
run "RT_NONE=0" "--emulate-world 8"
for rf in 8 24; do run "RT_REFILL=$rf" "--emulate-world 8"; done
for sm in 4 12; do run "RT_STEPMIN=$sm" "--emulate-world 8"; done
for pa in 8 24; do run "RT_PAIRAGAIN=$pa" "--emulate-world 8"; done
run "RT_NONE=0" "--emulate-world 8"
echo "== -mllvm -amdgpu-sched-strategy=max-memory-clause (_v/mc) against the default scheduler (.)"
bash profiles/r04_bisect.sh ". _v/mc"
bash profiles/r04_bisect.sh ". _v/mc" --emulate-world 8 --steps 12
bash profiles/r04_bisect.sh ". _v/mc" --workload config5 --steps 2 --warmup 1
} > gpurun_out/r04_batch4.txt 2>&1
cat gpurun_out/r04_batch4.txt
