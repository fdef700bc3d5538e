#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2_pow_24 or wide_table or partitions" > gpurun_out/region_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|rror" gpurun_out/region_pytest.log | tail -3
python tools/dist_ab.py 50000 centroflye_amd/libcfhip.so centroflye_amd/libcfhip.so:dist_regions=1 centroflye_amd/libcfhip.so:dist_regions=4 centroflye_amd/libcfhip.so:dist_wide=1 2>&1 | grep -v "^$" | tail -5
