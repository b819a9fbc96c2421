#!/bin/bash
# NOTE (round 5): this script sweeps RT_REFILL / RT_STEPMIN / RT_PAIRAGAIN / RT_DRAIN_LANES through the ENVIRONMENT, as the library read them when it was
# written.  They are compile-time constants since round 4 (csrc/rt_scene_dev.h; the library prints a warning when it sees one set): re-running
# it as it is gives identical rows.  A sweep is a rebuild per value now: profiles/bisect.sh over trees built with make EXTRA=-DRT_...=N.
# Round 4: the drain loop (trace_persistent, RT_DRAIN_LANES) A/B on ONE box.  bash profiles/r04_drain_ab.sh  -> gpurun_out/r04_ab_drain_loop.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_ab_drain_loop.txt
: > $OUT
echo "== lone step (profiles/lone_step.py), RT_DRAIN_LANES=0 | 1 | 4" >> $OUT
for k in 0 1 4; do echo "-- RT_DRAIN_LANES=$k" >> $OUT; RT_DRAIN_LANES=$k timeout -k 10 200 python3 profiles/lone_step.py 2>&1 | tail -4 >> $OUT; done
echo "== full frame (config 3, 1080p x 64 spp), one kernel at a time" >> $OUT
bash profiles/ab_bench.sh "RT_DRAIN_LANES=0" "RT_DRAIN_LANES=1" "RT_DRAIN_LANES=2" "RT_DRAIN_LANES=4" "RT_DRAIN_LANES=8" "RT_DRAIN_LANES=16" "RT_DRAIN_LANES=64" "RT_DRAIN_LANES=0" "RT_DRAIN_LANES=4" >> $OUT 2>&1
echo "== 1/8 share (--emulate-world 8)" >> $OUT
BENCH_ARGS="--emulate-world 8" STEPS=10 bash profiles/ab_bench.sh "RT_DRAIN_LANES=0" "RT_DRAIN_LANES=1" "RT_DRAIN_LANES=2" "RT_DRAIN_LANES=4" "RT_DRAIN_LANES=8" "RT_DRAIN_LANES=16" "RT_DRAIN_LANES=64" "RT_DRAIN_LANES=0" "RT_DRAIN_LANES=4" >> $OUT 2>&1
echo "== 1/8 share, the second stream: RT_FUSE=0 serial | 2 both streams at once | 3 gated (the gate really waits since round 4)" >> $OUT
BENCH_ARGS="--emulate-world 8" STEPS=10 bash profiles/ab_bench.sh "RT_FUSE=0" "RT_FUSE=2" "RT_FUSE=3" "RT_FUSE=0" "RT_FUSE=2" "RT_FUSE=3" >> $OUT 2>&1
echo "== --spp 16 (33 M samples), the second stream" >> $OUT
BENCH_ARGS="--spp 16" STEPS=10 bash profiles/ab_bench.sh "RT_FUSE=0" "RT_FUSE=2" "RT_FUSE=3" "RT_FUSE=0" "RT_FUSE=2" "RT_FUSE=3" >> $OUT 2>&1
echo "== --qlearn 8 (16.6 M-sample batches), the second stream" >> $OUT
BENCH_ARGS="--qlearn 8" STEPS=4 bash profiles/ab_bench.sh "RT_FUSE=0" "RT_FUSE=2" "RT_FUSE=3" >> $OUT 2>&1
echo "== Tick times (profiles/tick_time.py)" >> $OUT
for k in 0 4; do echo "-- RT_DRAIN_LANES=$k" >> $OUT; RT_DRAIN_LANES=$k timeout -k 10 200 python3 profiles/tick_time.py 2>&1 | tail -12 >> $OUT; done
cat $OUT
