#!/bin/bash
# Same-box A/B of the drain's helpers (RT_SPECULATE): the tree as it is, _v/pairs0 (the new drain loop, nothing handed out), _v/nospec (the plain
# per-lane drain); 1/8 share and full frame; then the tail probe builds with the helpers' tallies.  Built before the call: profiles/mkvariant.sh.
mkdir -p gpurun_out/r06
timeout -k 10 120 python -m pytest tests -m gpu -x -q -k "find_nearest_and_occlusion or edge_cases or golden or path_frames" > gpurun_out/r06/spec_t3.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/r06/spec_t3.log
[ $rc -eq 0 ] || exit 1
export TMO=120
STEPS=10 WARMUP=3 REPS=2 bash profiles/bisect.sh ". _v/pairs0 _v/nospec" --emulate-world 8 > gpurun_out/r06/spec_ab3.txt 2>&1
STEPS=10 WARMUP=3 REPS=1 bash profiles/bisect.sh ". _v/pairs0 _v/nospec" >> gpurun_out/r06/spec_ab3.txt 2>&1; cat gpurun_out/r06/spec_ab3.txt
for v in tp_spec tp_nospec; do (cd _v/$v && RT_FUSE=0 timeout -k 10 120 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-count --no-legs --emulate-world 8 > ../../gpurun_out/r06/tail3_$v.raw 2>&1; grep "tail probe" ../../gpurun_out/r06/tail3_$v.raw | tail -10 > ../../gpurun_out/r06/tail3_$v.txt); echo $v; cat gpurun_out/r06/tail3_$v.txt; done
