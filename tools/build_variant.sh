#!/usr/bin/env bash
# Developer tool: build a variant of libcfhip into centroflye_amd/build_variants/<name>.so
# usage: tools/build_variant.sh <name> [-DFLAG ...] [--dist <alternative cf_dist.hip>]
# (centroflye_amd/build_variants/ is listed in .gpurunignore so that left-over builds do not travel with every GPU lease: take the line out
#  for the A/B runs of tools/dist_ab.py / count_ab.py / dist_ablation.sh, delete the builds afterwards)
set -euo pipefail
root="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
flags=(); dist="$root/centroflye_amd/csrc/hip/cf_dist.hip"
while [[ $# -gt 0 ]]; do
  if [[ "$1" == "--dist" ]]; then dist=$2; shift 2; else flags+=("$1"); shift; fi
done
out="$root/centroflye_amd/build_variants"; mkdir -p "$out"
srcs=()
for f in "$root"/centroflye_amd/csrc/hip/*.hip; do [[ "$(basename "$f")" == cf_dist.hip ]] || srcs+=("$f"); done
srcs+=("$dist")
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -amdgpu-atomic-optimizer-strategy=None "${flags[@]}" -I"$root/include" -I"$root/centroflye_amd/csrc/hip" -shared -o "$out/$name.so" "${srcs[@]}" -ldl
echo "$out/$name.so"
