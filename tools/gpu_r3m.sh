#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $L $L > gpurun_out/r3m_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3m_ab.log
bash tools/gpu_stats.sh r3m | head -24
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3m_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3m_pytest.log
