#!/bin/bash
# The eight 1/8 shares of the configs BASELINE assigns to 8 GPUs, each rendered alone on ONE GPU
# (bench.py --emulate-world 8 --emulate-rank r), beside the whole step: max / mean share, projected speed-up before the exchange.
#   bash profiles/r05_shares.sh  -> gpurun_out/r05/shares_config4_config5.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/shares_config4_config5.txt
: > $OUT
one() { timeout -k 10 900 python3 bench.py --no-cpu-baseline --no-count "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; }
for spec in "config4|--workload config4 --steps 3 --warmup 1" "config5|--workload config5 --steps 2 --warmup 1" "config5 --qlearn 32|--workload config5 --qlearn 32 --steps 2 --warmup 1" "config3|--steps 8 --warmup 2"; do
  name=${spec%%|*}; args=${spec#*|}
  full=$(one $args)
  shares=""
  for r in 0 1 2 3 4 5 6 7; do shares="$shares $(one $args --emulate-world 8 --emulate-rank $r)"; done
  python3 - "$name" "$full" $shares >> $OUT <<'PY'
import sys
name, full, sh = sys.argv[1], float(sys.argv[2]), [float(x) for x in sys.argv[3:]]
print("%-22s world 1: %9.3f ms | ranks 0..7 = %s | max %.3f mean %.3f spread %.1f %% | projected speed-up before the exchange: %.2fx (full / max), efficiency %.1f %%" % (
    name, full, " ".join("%.3f" % x for x in sh), max(sh), sum(sh) / len(sh), 100 * (max(sh) - min(sh)) / (sum(sh) / len(sh)), full / max(sh), 100 * full / max(sh) / 8))
PY
  tail -1 $OUT
done
