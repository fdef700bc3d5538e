#!/usr/bin/env bash
mkdir -p gpurun_out
python3 tools/c2_stamps.py 50000 > gpurun_out/r2m.log 2>&1; cat gpurun_out/r2m.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r2m_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2m_pytest.log
