#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $L $V/lb640.so:dist_wgs=2,dist_block=640 $V/lb768.so:dist_wgs=2,dist_block=768 $V/lb640.so $L:dist_wgs=2,dist_block=448 $L:dist_wgs=2,dist_block=576 > gpurun_out/r3l_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3l_ab.log
