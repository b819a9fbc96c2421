#!/bin/bash
# A short set of PMC passes (VALU, HBM bytes, clocks) of the bench, one kernel at a time, summarised per kernel:
#   bash profiles/pmc_quick.sh <tag> [ENV=..]...     -> profiles/<tag>_pmc_quick.json (+ raw under gpurun_out/)
set -u
TAG=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
export RT_FUSE=0 "$@"
OUT=gpurun_out/${TAG}_pmcq
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 2 --warmup 0 --no-cpu-baseline --no-count ${BENCH_ARGS:-}"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 profiles/pmc_summary.py $OUT > gpurun_out/${TAG}_pmcq_summary.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_pmcq_summary.json"))
out = {}
for k, v in sorted(d.items()):
    c = v["counters"]
    ms = v["ms_by_pass"].get("sq1", 0)
    if "GRBM_GUI_ACTIVE" not in c or ms <= 0: continue
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    hbm = c.get("FETCH_SIZE", 0) * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024
    o = {"launches": v["launches"], "ms": ms, "ms_per_step": round(ms / 2, 3),
         "lanes_enabled": round(c["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64.0 * c["SQ_ACTIVE_INST_VALU"]), 3),
         "valu_pipe_busy": round(2.0 * c["SQ_INSTS_VALU"] / (1024 * cycles), 3),
         "frac_of_peak_lane_ops": round(c["SQ_THREAD_CYCLES_VALU"] / (78.6432e12 * ms * 1e-3), 4),
         "lane_ops_per_step_G": round(c["SQ_THREAD_CYCLES_VALU"] / 2 / 1e9, 2),
         "wave_wait_frac": round(c["SQ_WAIT_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"]), 3),
         "hbm_GB_per_step": round(hbm / 2 / 1e9, 3), "hbm_TBps": round(hbm / (ms * 1e-3) / 1e12, 3),
         "fetch_GB_per_step": round(c.get("FETCH_SIZE", 0) * 1024 * 2 / 2 / 1e9, 3), "write_GB_per_step": round(c.get("WRITE_SIZE", 0) * 1024 / 2 / 1e9, 3),
         "l2_hit": round(c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_REQ_sum", 0)), 3)}
    out[k] = o
    print("%-36s %s" % (k, o))
json.dump(out, open("profiles/${TAG}_pmc_quick.json", "w"), indent=1)
PY
cp profiles/${TAG}_pmc_quick.json gpurun_out/
