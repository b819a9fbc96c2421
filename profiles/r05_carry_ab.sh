#!/bin/bash
# Carry A/B on one box: full step, the 1/8 share, the path Tick; RT_CARRY=0 is the round-4 round loop.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/carry_ab.txt
: > $OUT
echo "== path Tick, 1920x1080, no host mirror (profiles/tick_path.py)" >> $OUT
for e in "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=0" "RT_CARRY=2 RT_CARRY_K=8" "RT_CARRY=2 RT_CARRY_K=16" "RT_CARRY=2 RT_CARRY_K=32" "RT_CARRY=2 RT_CARRY_K=64" "RT_CARRY=1 RT_CARRY_K=16" "RT_CARRY=3 RT_CARRY_K=16" "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=16 RT_FUSE=2" "RT_CARRY=0 RT_FUSE=2" "RT_CARRY=2 RT_CARRY_K=16 RT_FUSE=0" "RT_CARRY=0 RT_FUSE=0"; do
  ( export $e; timeout -k 10 200 python3 profiles/tick_path.py 2>/dev/null >> $OUT )
done
echo "== 1/8 share (--emulate-world 8 --emulate-rank 4)" >> $OUT
BENCH_ARGS="--emulate-world 8 --emulate-rank 4" STEPS=8 bash profiles/ab_bench.sh "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=0" "RT_CARRY=2 RT_CARRY_K=8" "RT_CARRY=2 RT_CARRY_K=16" "RT_CARRY=2 RT_CARRY_K=32" "RT_CARRY=2 RT_CARRY_K=64" "RT_CARRY=1 RT_CARRY_K=16" "RT_CARRY=3 RT_CARRY_K=16" "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=16" >> $OUT
echo "== full step (config 3, 1080p x 64 spp)" >> $OUT
STEPS=6 bash profiles/ab_bench.sh "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=0" "RT_CARRY=2 RT_CARRY_K=16" "RT_CARRY=2 RT_CARRY_K=64" "RT_CARRY=0" "RT_CARRY=2 RT_CARRY_K=16" >> $OUT
cat $OUT
