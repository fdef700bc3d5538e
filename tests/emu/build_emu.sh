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
# one compiler process per source, as many at a time as there are cores (the kernels of cf_dist.hip alone are a third of the build)
obj="$here/obj_$(basename "$out" .so).$$"; mkdir -p "$obj"; trap 'rm -rf "$obj"' EXIT
printf '%s\n' "${srcs[@]}" | xargs -P "$(nproc)" -I{} bash -c 'g++ "$@" -c -x c++ "$0" -o "'"$obj"'/$(basename "$0").o"' {} \
    "${opt[@]}" "${san[@]}" -g -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-sign-compare -Wno-attributes \
    -I"$here" -I"$root/include" -I"$root/centroflye_amd/csrc/hip"
g++ "${san[@]}" -shared -o "$out.tmp$$" "$obj"/*.o
mv "$out.tmp$$" "$out"
