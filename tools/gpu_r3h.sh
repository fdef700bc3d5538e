#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $V/r2.so $L $V/r2.so $L > gpurun_out/r3h_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3h_ab.log
python3 tools/dist_stamps.py 50000 2>&1 | tail -2
