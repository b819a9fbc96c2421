#!/bin/bash
# Time the bench workload under different environment settings on one box: sweep_env.sh "A=1 B=2" "A=3" ...  (first and last run: defaults)
run() { echo "[$1]: $(env $1 python bench.py --no-cpu-baseline --steps 4 --warmup 1 $BENCH_ARGS 2>/dev/null | grep -o 'ms_per_step.: [0-9.]*\|kernel_ms_per_step.: {[^}]*}' | tr '\n' ' ')"; }
run "RT_NONE=0"
for v in "$@"; do run "$v"; done
run "RT_NONE=0"
