#!/usr/bin/env bash
# Build the HOST-EMULATED device library (test infrastructure only; see hip/hip_runtime.h).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/libcfhip_emu.so"
srcs=("$root"/centroflye_amd/csrc/hip/*.hip "$here/cfemu_runtime.cpp")
newest=$(ls -t "${srcs[@]}" "$root"/centroflye_amd/csrc/hip/*.h "$root"/include/cfhip.h "$here/hip/hip_runtime.h" | head -1)
if [[ -f "$out" && "$out" -nt "$newest" ]]; then exit 0; fi
g++ -O2 -g -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-sign-compare \
    -I"$here" -I"$root/include" -I"$root/centroflye_amd/csrc/hip" \
    -x c++ "${srcs[@]}" -o "$out.tmp$$"
mv "$out.tmp$$" "$out"
