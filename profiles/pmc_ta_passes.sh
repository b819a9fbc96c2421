#!/bin/bash
# Extra PMC passes on the vector-memory path (texture addresser / data, L1 stalls) for the traversal kernels.
# Usage: profiles/pmc_ta_passes.sh <outdir> [bench args...]; summary with profiles/pmc_summary.py <outdir>
set -u
OUT=$1; shift
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
BENCH_ARGS="${*:---steps 1 --warmup 0 --no-cpu-baseline}"
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
run ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum
run ta2 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
run td1 TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum
run tcp3 TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
run sq3 SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run sq4 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL
run sq5 SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VSKIPPED
