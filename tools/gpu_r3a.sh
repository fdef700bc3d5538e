#!/bin/bash
# round 3, first GPU trip: 64/64 CPU parity run of the bench workload, the new full-size tests, the bench line
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
nproc; free -g | head -2
timeout 1500 python3 tools/full_parity.py > gpurun_out/full_parity.log 2>&1; echo "full_parity rc=$?"; tail -2 gpurun_out/full_parity.log | cut -c1-600
[ -f gpurun_out/r03_full_parity.json ] && cp gpurun_out/r03_full_parity.json profiles/r03_full_parity.json
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --durations=8 > gpurun_out/r3a_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3a_pytest.log
timeout 900 python3 bench.py > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3a_bench.err; cut -c1-1500 gpurun_out/r3a_bench.json
