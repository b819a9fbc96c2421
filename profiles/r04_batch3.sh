#!/bin/bash
# Round 4, third measurement batch on one box: the ordered traversal queue (RT_ORDER), eight waves per SIMD, the Q-learning shade
# kernel at four waves, one launch per round at the 1/8 share.  bash profiles/r04_batch3.sh -> gpurun_out/r04_batch3.txt
cd "$(dirname "$0")/.."
{
echo "== RT_ORDER, config 3"; bash profiles/r04_bisect.sh ".@RT_ORDER=0 .@RT_ORDER=1"
echo "== RT_ORDER, config 4"; bash profiles/r04_bisect.sh ".@RT_ORDER=0 .@RT_ORDER=1" --workload config4 --steps 4
echo "== RT_ORDER, config 5"; bash profiles/r04_bisect.sh ".@RT_ORDER=0 .@RT_ORDER=1" --workload config5 --steps 2 --warmup 1
echo "== RT_ORDER, 1/8 share"; bash profiles/r04_bisect.sh ".@RT_ORDER=0 .@RT_ORDER=1" --emulate-world 8 --steps 12
echo "== eight waves per SIMD (20 KB of LDS per block), config 3 / 4 / 5 / 1/8 share / config 2"
bash profiles/r04_bisect.sh ". _v/w8"
bash profiles/r04_bisect.sh ". _v/w8" --workload config4 --steps 4
bash profiles/r04_bisect.sh ". _v/w8" --workload config5 --steps 2 --warmup 1
bash profiles/r04_bisect.sh ". _v/w8" --emulate-world 8 --steps 12
bash profiles/r04_bisect.sh ". _v/w8" --workload config2
echo "== 1/8 share: one traversal launch per round (RT_FUSE=1) against two streams (default)"
bash profiles/r04_bisect.sh ".@RT_FUSE=2 .@RT_FUSE=1 _v/w8@RT_FUSE=1" --emulate-world 8 --steps 12
echo "== Q-learning shade kernel at 5 waves (scratch) against 4 waves (none): config 5 --qlearn 32, config 3 --qlearn 8"
bash profiles/r04_bisect.sh ". _v/q4" --workload config5 --qlearn 32 --steps 2 --warmup 1
bash profiles/r04_bisect.sh ". _v/q4" --qlearn 8 --steps 4
} > gpurun_out/r04_batch3.txt 2>&1
cat gpurun_out/r04_batch3.txt
