#!/bin/bash
# The bench lines kept under profiles/ for a round: bash profiles/bench_set.sh <tag>   (GPU box; writes gpurun_out/<tag>_bench*.json)
set -u
TAG=$1
cd "$(dirname "$0")/.."
run() { name=$1; shift; timeout -k 10 ${TMO:-400} python3 bench.py "$@" 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_bench${name}.json; python3 -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench${name}.json')); print('${name:-_default}', d['ms_per_step'], d['value'], d['frame_checksum'], d['roofline'].get('frac'), d['roofline']['kernel_ms_per_step'])"; }
run "" --steps 10 --warmup 3
run _serial_nocpu --steps 10 --warmup 3 --no-cpu-baseline
for s in 8 16 32; do run _spp$s --steps 10 --warmup 3 --spp $s --no-cpu-baseline; done
for w in 2 4 8; do run _emulate_world$w --steps 10 --warmup 3 --emulate-world $w --no-cpu-baseline; done
run _config2 --steps 10 --warmup 3 --workload config2 --no-cpu-baseline
run _config4 --steps 5 --warmup 2 --workload config4 --no-cpu-baseline
run _config5 --steps 2 --warmup 1 --workload config5 --no-cpu-baseline
run _config5_qlearn --steps 2 --warmup 1 --workload config5 --qlearn 32 --no-cpu-baseline
for r in 0 1 2 3 4 5 6 7; do run _emulate_world8_rank$r --steps 10 --warmup 3 --emulate-world 8 --emulate-rank $r --no-cpu-baseline; done
run _config3_qlearn --steps 3 --warmup 1 --qlearn 8 --no-cpu-baseline
