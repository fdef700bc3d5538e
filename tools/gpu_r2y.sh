#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2_pow_24 or beyond_255 or wide_table or stage2_fixture" > gpurun_out/r2y_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2y_pytest.log
python tools/dist_ab.py 50000 centroflye_amd/libcfhip.so centroflye_amd/libcfhip.so:dist_dbits=7 centroflye_amd/libcfhip.so:dist_dbits=6 2>&1 | grep -v "^$" | tail -6
