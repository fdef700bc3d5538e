#!/usr/bin/env bash
# Build the HOST-EMULATED device library (test infrastructure only; see hip/hip_runtime.h).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
out="$here/libcfhip_emu.so"
san=()
opt=(-O2)
# CF_EMU_UBSAN=1: the same sources with UndefinedBehaviorSanitizer (tests/test_emu_ubsan.py) -> libcfhip_emu_ubsan.so
if [[ "${CF_EMU_UBSAN:-0}" == 1 ]]; then out="$here/libcfhip_emu_ubsan.so"; san=(-fsanitize=undefined -fno-sanitize-recover=undefined -fno-sanitize=alignment); opt=(-O1); fi
# every kernel source except the RCCL transport; its place is taken by the file-based transport of the emulator
srcs=()
for f in "$root"/centroflye_amd/csrc/hip/*.hip; do [[ "$(basename "$f")" == cf_comm_rccl.hip ]] || srcs+=("$f"); done
srcs+=("$here/cfemu_runtime.cpp" "$here/cf_comm_emu.cpp")
newest=$(ls -t "${srcs[@]}" "$root"/centroflye_amd/csrc/hip/*.h "$root"/include/cfhip.h "$here/hip/hip_runtime.h" | head -1)
if [[ -f "$out" && "$out" -nt "$newest" ]]; then exit 0; fi
g++ "${opt[@]}" "${san[@]}" -g -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-sign-compare \
    -I"$here" -I"$root/include" -I"$root/centroflye_amd/csrc/hip" \
    -x c++ "${srcs[@]}" -o "$out.tmp$$"
mv "$out.tmp$$" "$out"
