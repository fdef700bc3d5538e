#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -x -q > gpurun_out/r2w_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2w_pytest.log
bash tools/gpu_cli_wall.sh 2>&1 | tail -22
