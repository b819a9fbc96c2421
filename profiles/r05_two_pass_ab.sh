#!/bin/bash
# Two-pass traversal launches (rt_stream.h "Two passes") A/B on one box: RT_CARRY_K = steps a ray still makes after its wave's queue ran dry
# before it is parked for the second pass; -1 = one pass (round 4's launches).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/two_pass_ab.txt
: > $OUT
echo "== path Tick, 1920x1080, no host mirror (profiles/tick_path.py)" >> $OUT
for k in -1 0 8 16 32 64 128 -1 32; do ( export RT_CARRY_K=$k; timeout -k 10 200 python3 profiles/tick_path.py 2>/dev/null >> $OUT ); done
echo "== 1/8 share (--emulate-world 8 --emulate-rank 4)" >> $OUT
BENCH_ARGS="--emulate-world 8 --emulate-rank 4" STEPS=8 bash profiles/ab_bench.sh "RT_CARRY_K=-1" "RT_CARRY_K=0" "RT_CARRY_K=8" "RT_CARRY_K=16" "RT_CARRY_K=32" "RT_CARRY_K=64" "RT_CARRY_K=128" "RT_CARRY_K=-1" "RT_CARRY_K=32" >> $OUT
echo "== full step (config 3, 1080p x 64 spp)" >> $OUT
STEPS=6 bash profiles/ab_bench.sh "RT_CARRY_K=-1" "RT_CARRY_K=16" "RT_CARRY_K=32" "RT_CARRY_K=64" "RT_CARRY_K=-1" "RT_CARRY_K=32" >> $OUT
cat $OUT
