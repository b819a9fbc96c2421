#!/bin/bash
# profiles/out_of_cache.py under rocprofv3: one plain run with the oracle check, then separate --pmc passes (counters only) -> gpurun_out/<tag>_out_of_cache*.json
#   bash profiles/out_of_cache.sh [n] [spp] [tag]      (GPU box, repository root; copy the two JSON files into profiles/ afterwards)
# spp 4 = 8.3 M samples per batch: one traversal launch per round (k_traverse_s); spp 16 = 33 M: k_extend_s and k_connect_s on two streams.
# The result is stamped with bench.py's kernel_hash of the sources it ran on: bench.py quotes it only while that is the tree's.
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
N=${1:-2048}
SPP=${2:-4}
TAG=${3:-r05}
OUT=gpurun_out/${TAG}_ooc_spp$SPP
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 profiles/out_of_cache.py --n $N --spp $SPP --check --json $OUT/run.json > $OUT/run.log 2>&1 || { echo "plain run failed"; tail -5 $OUT/run.log; exit 1; }
tail -2 $OUT/run.log | head -1
run() { name=$1; shift; timeout -k 10 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 profiles/out_of_cache.py --n $N --spp $SPP --steps 2 > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_BUSY_avr TA_BUSY_max
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq3 SQ_INSTS_VMEM SQ_INSTS_VALU
python3 profiles/pmc_summary.py $OUT > gpurun_out/${TAG}_out_of_cache_spp${SPP}_pmc_summary.json
python3 - $OUT gpurun_out/${TAG}_out_of_cache_spp${SPP} <<PY
import json, sys
sys.path.insert(0, '.')
import bench
d = json.load(open(sys.argv[2] + "_pmc_summary.json"))
run = json.load(open(sys.argv[1] + "/run.json"))
out = {"kernel_hash": bench.kernel_hash(run["build"].split(" | ")[0]), "run": run, "kernels": {}}
for k, v in sorted(d.items()):
    c = v["counters"]; ms = v["ms_by_pass"].get("fetch", 0); n = v["launches"]
    if ms <= 0: continue
    # HBM bytes as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts 64-byte... units of KiB here, doubled (the guide's correction); WRITE_SIZE in KiB
    hbm = c.get("FETCH_SIZE", 0) * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    o = {"launches": n, "ms_total": ms, "hbm_GB": round(hbm / 1e9, 3), "hbm_TBps": round(hbm / (ms * 1e-3) / 1e12, 3),
         "frac_of_8TBps": round(hbm / (ms * 1e-3) / 8e12, 4), "frac_of_6.29TBps_measured_peak": round(hbm / (ms * 1e-3) / 6.29e12, 4),
         "l2_hit_rate": round(c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0)), 4),
         "ta_busy_avg": round(c.get("TA_BUSY_avr", 0) / max(cyc, 1), 4), "ta_busy_max": round(c.get("TA_BUSY_max", 0) / max(cyc, 1), 4),
         "lanes_enabled": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64.0 * c.get("SQ_ACTIVE_INST_VALU", 0)), 3),
         "valu_pipe_busy": round(2.0 * c.get("SQ_INSTS_VALU", 0) / max(1.0, 1024 * cyc), 3) if "sq1" in v["ms_by_pass"] else None,
         "wave_wait_frac": round(c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 0)), 3),
         "vmem_insts_M": round(c.get("SQ_INSTS_VMEM", 0) / 1e6, 1)}
    out["kernels"][k] = o
    print("%-40s %s" % (k, o))
json.dump(out, open(sys.argv[2] + ".json", "w"), indent=1)
PY
find $OUT -name "*.csv" -delete
