#!/bin/bash
# Every rank's share of an N-rank step by itself on ONE GPU (bench.py --emulate-world N --emulate-rank r): the N-GPU step is as long
# as its SLOWEST share, so the projected speed-up is (1-GPU step) / max, not / rank 0.  bash profiles/r04_shares.sh -> gpurun_out/r04_shares_all_ranks.txt
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_shares_all_ranks.txt
one() { timeout -k 10 300 python3 bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f' % d['ms_per_step'])"; }
{
echo "bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-count, config 3 (1920x1080 x 64 spp), ms per step, one box"
full=$(one); echo "world 1: $full"
for N in 2 4 8; do
  line=""; max=0; sum=0
  for r in $(seq 0 $((N-1))); do t=$(one --emulate-world $N --emulate-rank $r); line="$line $t"; max=$(python3 -c "print(max($max,$t))"); sum=$(python3 -c "print($sum+$t)"); done
  python3 -c "
n=$N; mx=$max; mean=$sum/$N; full=$full
print('world %d: ranks 0..%d =%s | max %.3f mean %.3f spread %.1f %% | projected speed-up before the gather: %.2fx (full / max), efficiency %.1f %%' % (n, n-1, '$line', mx, mean, 100*(mx/mean-1), full/mx, 100*full/mx/n))"
done
} | tee $OUT
