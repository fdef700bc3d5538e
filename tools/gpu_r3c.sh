#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 300 ./tools/ubench/issue_mix > gpurun_out/r3c_ubench.log 2>&1; echo "ubench rc=$?"; cat gpurun_out/r3c_ubench.log
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $V/r2.so $V/pfa1.so $L $V/pfa3.so $V/pfa2b2.so $L:dist_wgs=4,dist_block=256 $L:dist_wgs=3,dist_block=320 $L:dist_wgs=2,dist_block=384 $L:dist_wgs=2,dist_block=256 $V/r2.so $L > gpurun_out/r3c_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3c_ab.log
timeout 900 python3 tools/place_check.py 50000 2 > gpurun_out/r3c_place_check.log 2>&1; echo "place_check rc=$?"; cat gpurun_out/r3c_place_check.log
