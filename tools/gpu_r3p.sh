#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=centroflye_amd/libcfhip.so
timeout 900 python3 tools/dist_ab.py 50000 $L $L > gpurun_out/r3p_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3p_ab.log
python3 tools/dist_stamps.py 50000 2>&1 | tail -2
timeout 1800 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config2_50k_reads_distance or config0" > gpurun_out/r3p_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3p_pytest.log
