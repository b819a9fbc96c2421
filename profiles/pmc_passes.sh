#!/bin/bash
# PMC passes for the traversal kernels (separate rocprofv3 runs, counters only: no --kernel-trace/--stats
# combined with traces that gpurun refuses).  Usage: profiles/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $BENCH_ARGS > $OUT/$name.log 2>&1 || echo "pass $name failed"; }
BENCH_ARGS="${*:---steps 1 --warmup 0 --spp 8 --no-cpu-baseline}"
mkdir -p $OUT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_GATE_EN1_sum
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
# the texture addresser (vector-memory address path) and what it is fed; TA_ADDR_STALLED_BY_TD / TD_* hang on this pool: not collected
run ta1 TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum
run sq3 SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_IFETCH
