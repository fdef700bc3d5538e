#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python3 tools/dist_ab.py 50000 centroflye_amd/build_variants/r2.so centroflye_amd/libcfhip.so centroflye_amd/build_variants/r2.so centroflye_amd/libcfhip.so > gpurun_out/r3b_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3b_ab.log
timeout 900 python3 tools/place_debug.py 50000 2 centroflye_amd/build_variants/r2.so centroflye_amd/build_variants/r2.so:place_fused=0 centroflye_amd/libcfhip.so > gpurun_out/r3b_place.log 2>&1; echo "place rc=$?"; cat gpurun_out/r3b_place.log
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --durations=8 > gpurun_out/r3b_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3b_pytest.log
timeout 900 python3 bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3b_bench.err; cut -c1-1200 gpurun_out/r3b_bench.json
