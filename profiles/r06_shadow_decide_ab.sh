#!/bin/bash
# Shadow rays answered where they are made (k_shade_s runs their first traversal step): parity subset, then same-box A/B against _v/base (HEAD before
# the change) on the full frame, the 1/8 share, configs 2 / 4 / 5.
mkdir -p gpurun_out/r06
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "path_frames or stream_pipeline or full_size or qlearning_sampler or bench" > gpurun_out/r06/sd_t.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/r06/sd_t.log
[ $rc -eq 0 ] || exit 1
export TMO=300
{ STEPS=10 WARMUP=3 REPS=2 bash profiles/bisect.sh ". _v/base"
  STEPS=10 WARMUP=3 REPS=2 bash profiles/bisect.sh ". _v/base" --emulate-world 8
  STEPS=6 WARMUP=2 REPS=1 bash profiles/bisect.sh ". _v/base" --workload config2
  STEPS=3 WARMUP=1 REPS=1 bash profiles/bisect.sh ". _v/base" --workload config4
  STEPS=2 WARMUP=1 REPS=1 bash profiles/bisect.sh ". _v/base" --workload config5
} 2>&1 | tee gpurun_out/r06/shadow_decide_ab.txt
timeout -k 10 120 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('rays per step', d['rays_per_step'], d['all_rays_mrays_per_s'])"
