#!/usr/bin/env python3
"""Roofline inputs for bench.py from a PMC summary (profiles/pmc_summary.py output) -> profiles/roofline_pmc.json.

For the traversal kernel k_extend (all timed variants summed) per step of the bench workload:
  lane_ops            SQ_THREAD_CYCLES_VALU: enabled-lane VALU instruction slots (one per lane per VALU instruction)
  valu_insts          SQ_INSTS_VALU (wave-level)
  lanes_enabled       SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  valu_pipe_busy      2 cycles * SQ_INSTS_VALU / (SIMDs * kernel cycles): a SIMD-32 retires a wave64 VALU instruction in
                      2 cycles when two or more waves feed it (MI355X_MICROARCH.md: v_fma_f32 2 cyc, one wave alone 4;
                      profiles/microbench/valu_rate.hip measures 2.0-2.2)
  wave_valu_active    4 cycles * SQ_ACTIVE_INST_VALU / (SIMDs * kernel cycles): the SQ's own per-wave view (a wave is
                      "VALU active" for 4 cycles per instruction; up to two waves of a SIMD can be at once)
  frac                lane_ops / (peak lane-ops/s * kernel seconds): <= 1 by construction; = lanes_enabled x pipe busy
  hbm_bytes           FETCH_SIZE x 2 (gfx950 correction of the guide's HBM section) + WRITE_SIZE, in bytes
  ta_busy_avg / max   TA_BUSY_avr / TA_BUSY_max over the kernel's cycles: the share of time the texture addressers (one per
                      CU: every vector-memory instruction passes through) are busy, averaged over them / for the busiest
  ta_cycles_per_vmem_inst   TA_TA_BUSY_sum / SQ_INSTS_VMEM: addresser-busy cycles per vector-memory instruction
The file is stamped with a hash of the kernel sources (csrc/); bench.py ignores it when the sources changed.
One counter set per number of ranks: the file holds {"by_world": {"1": {...}, "2": {...}, ...}}; the set for N ranks is measured on
ONE GPU with bench.py --emulate-world N (rank 0's rows of the N-rank shard), so that rank 0 of an N-GPU run is priced with the
lane-ops of its own share.
Usage: valu_roofline.py <pmc_summary.json> <workload> <width> <height> <spp> <kernel_hash of the profiled run's bench line> [world]"""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # CUs x SIMD-32 x lanes x 2.4 GHz = 78.6 T lane-ops/s (= 157.3 TFLOP/s fp32 FMA)
SIMDS = 1024


def kernel_hash():
    """bench.py's kernel_hash(): sources + the library's build / tuning account, read from the bench line of the profiled run
    (its roofline.pmc.kernel_hash), passed as argv[6] -- a hash recomputed here could not know the environment of that run."""
    return sys.argv[6]


def main():
    src = sys.argv[1]
    d = json.load(open(src))
    world = int(sys.argv[7]) if len(sys.argv) > 7 else 1
    out = {"kernel_hash": kernel_hash(), "world": world, "source": os.path.relpath(os.path.abspath(src), ROOT), "workload": [sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])],
           "peak_lane_ops_per_s": PEAK_LANE_OPS, "kernels": {}}
    groups = {"k_extend": [k for k in d if k.startswith("k_extend")], "k_connect": [k for k in d if k.startswith("k_connect")],
              "k_shade": [k for k in d if k.startswith("k_shade")]}
    for g, ks in groups.items():
        if not ks:
            continue
        c = {}
        launches, ms = 0, 0.0
        for k in ks:
            launches += d[k]["launches"]
            ms += d[k]["ms_by_pass"].get("sq1", list(d[k]["ms_by_pass"].values())[0])
            for n, v in d[k]["counters"].items():
                c[n] = c.get(n, 0.0) + v
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        sec = ms * 1e-3
        o = {"launches": launches, "ms": round(ms, 3), "clock_ghz": round(cycles / sec / 1e9, 3),
             "lane_ops": c["SQ_THREAD_CYCLES_VALU"], "valu_insts": c["SQ_INSTS_VALU"], "salu_insts": c.get("SQ_INSTS_SALU"),
             "lanes_enabled": round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4),
             "valu_pipe_busy": round(2.0 * c["SQ_INSTS_VALU"] / (SIMDS * cycles), 4),
             "wave_valu_active": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cycles), 4),
             "frac_of_peak_lane_ops": round(c["SQ_THREAD_CYCLES_VALU"] / (PEAK_LANE_OPS * sec), 4),
             "wave_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4),
             "l1_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "l2_requests": c.get("TCP_TCC_READ_REQ_sum"),
             "l1_accesses_per_cu_cycle": round(c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / (256 * cycles), 4),
             "l2_hit_rate": round(c["TCC_HIT_sum"] / max(1.0, c["TCC_REQ_sum"]), 4),
             "hbm_bytes": int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)}
        if "TA_BUSY_avr" in c and cycles > 0:
            o["ta_busy_avg"] = round(c["TA_BUSY_avr"] / cycles, 4)
            o["ta_busy_max"] = round(c["TA_BUSY_max"] / cycles, 4)
            o["vmem_insts"] = c.get("SQ_INSTS_VMEM")
            if c.get("SQ_INSTS_VMEM"):
                o["ta_cycles_per_vmem_inst"] = round(c["TA_TA_BUSY_sum"] / c["SQ_INSTS_VMEM"], 2)
        o["hbm_bytes_per_launch"] = int(o["hbm_bytes"] / launches)
        o["lane_ops_per_launch"] = o["lane_ops"] / launches
        out["kernels"][g] = o
    path = os.environ.get("RT_ROOFLINE_PMC_OUT") or os.path.join(ROOT, "profiles", "roofline_pmc.json")
    try:
        allw = json.load(open(path))
        if "by_world" not in allw:
            allw = {"by_world": {}}
    except Exception:
        allw = {"by_world": {}}
    allw["by_world"][str(world)] = out
    json.dump(allw, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
