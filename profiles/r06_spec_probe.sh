#!/bin/bash
# Tail probe of the 1/8 share (and, FULL=1, of the full frame) for a list of variant trees under _v/: bash profiles/r06_spec_probe.sh "tp_spec_ns tp_nospec ..."
mkdir -p gpurun_out/r06
for v in $1; do
  (cd _v/$v && RT_FUSE=0 timeout -k 10 120 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-count --no-legs --emulate-world 8 > ../../gpurun_out/r06/probe_$v.raw 2>&1; grep "tail probe" ../../gpurun_out/r06/probe_$v.raw | tail -10 > ../../gpurun_out/r06/probe_$v.txt)
  echo "== $v (1/8 share)"; cut -c1-200 gpurun_out/r06/probe_$v.txt
  if [ -n "$FULL" ]; then
    (cd _v/$v && RT_FUSE=0 timeout -k 10 120 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-count --no-legs > ../../gpurun_out/r06/probe_full_$v.raw 2>&1; grep "tail probe" ../../gpurun_out/r06/probe_full_$v.raw | tail -10 > ../../gpurun_out/r06/probe_full_$v.txt)
    echo "== $v (full frame)"; cut -c1-200 gpurun_out/r06/probe_full_$v.txt
  fi
done
