#!/bin/bash
# The round's final records, after profiles/roofline_passes.sh r06_final "1 2 4 8" (whose roofline_pmc.json is in profiles/ already):
# out-of-cache counters -> profiles/out_of_cache_pmc.json, then the driver's bench command -> gpurun_out/r06/final_bench.json, then the
# per-kernel counters and the bench set.
set -u
mkdir -p gpurun_out/r06; export TMPDIR=/tmp
bash profiles/out_of_cache.sh 2048 16 r06 > gpurun_out/r06/out_of_cache.log 2>&1; echo "out_of_cache rc=$?"
cp gpurun_out/r06_out_of_cache_spp16.json profiles/out_of_cache_pmc.json
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/final_bench.json 2> gpurun_out/r06/final_bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('gpurun_out/r06/final_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; o=r['hbm']['out_of_cache_live']
print('ms', d['ms_per_step'], 'frac', r['frac'], 'sides', {k:(v if not isinstance(v,dict) else v.get('live')) for k,v in (r['sides'] or {}).items() if k in ('ta_busy','l1_accesses_per_cu_cycle','valu_enabled_lane_frac_of_peak','hbm_counter_frac_of_peak','hbm_algorithmic_demand_over_peak')})
print('ooc', o['kernel_ms'], o['algorithmic_over_hbm_peak'], o['frac_of_hbm_peak'], o['crop_parity'])
print('share', d['share_ms']['slowest'], d['share_ms']['projected_speedup'], 'tick', d['tick_ms']['whitted'], d['tick_ms']['path'], 'legs', d['gpu_leg_s'], d['cpu_leg_s'])
"
bash profiles/pmc_quick.sh r06_final > gpurun_out/r06/pmc_quick.log 2>&1; echo "pmc_quick rc=$?"
TMO=300 bash profiles/bench_set.sh r06_final > gpurun_out/r06_final_bench_set.txt 2>&1; echo "bench_set rc=$?"; cat gpurun_out/r06_final_bench_set.txt
