#!/bin/bash
# The traversal queue with the rays off specular surfaces first (CL_LONG) against entry order (_v/nolong: -DRT_LONG_FIRST=0): parity subset, A/B on the
# 1/8 share, the full frame, configs 2 / 4 / 5, Tick.
mkdir -p gpurun_out/r06
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "path_frames or stream_pipeline or full_size or bench or tick" > gpurun_out/r06/lf_t.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r06/lf_t.log
[ $rc -eq 0 ] || exit 1
export TMO=300
{ STEPS=10 WARMUP=3 REPS=3 bash profiles/bisect.sh ". _v/nolong" --emulate-world 8
  STEPS=10 WARMUP=3 REPS=2 bash profiles/bisect.sh ". _v/nolong"
  STEPS=6 WARMUP=2 REPS=1 bash profiles/bisect.sh ". _v/nolong" --workload config2
  STEPS=3 WARMUP=1 REPS=1 bash profiles/bisect.sh ". _v/nolong" --workload config4
  STEPS=2 WARMUP=1 REPS=1 bash profiles/bisect.sh ". _v/nolong" --workload config5
} 2>&1 | tee gpurun_out/r06/long_first_ab.txt
