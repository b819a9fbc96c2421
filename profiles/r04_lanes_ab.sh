#!/bin/bash
# A/B of RT_LANES (one or two pipelines per batch of the dense path) on one box: the full bench step and the shares of 2, 4, 8 ranks
cd "$(dirname "$0")/.."
for w in 0 2 4 8; do
  args=""; [ $w != 0 ] && args="--emulate-world $w"
  echo "== bench.py $args"
  bash profiles/r04_bisect.sh "./@RT_LANES=1 ./@RT_LANES=2" $args || exit 1
done
