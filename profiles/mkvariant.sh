#!/bin/bash
# A copy of the product tree under _v/<name>, built with make EXTRA="<flags>": the other side of a same-box A/B (profiles/bisect.sh).
#   bash profiles/mkvariant.sh nospec "-DRT_SPECULATE=0"         (_v/ is git-ignored and travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; extra=${2:-}
rm -rf _v/$name; mkdir -p _v/$name/profiles
cp -r bench.py include ray-and-pathtracer_amd _v/$name/
cp profiles/roofline_pmc.json _v/$name/profiles/ 2>/dev/null || true
find _v/$name -name "*.so" -delete; find _v/$name -name "__pycache__" -prune -exec rm -rf {} +
make -s -C _v/$name/ray-and-pathtracer_amd/csrc EXTRA="$extra" 2>&1 | grep -E "error" || true
make -s -C _v/$name/ray-and-pathtracer_amd/host 2>&1 | grep -E "error" || true
ls -la _v/$name/ray-and-pathtracer_amd/csrc/librt_amd.so
