#!/bin/bash
# GPU box: the complete -m gpu suite, __graft_entry__.smoke() and one default bench.py run (logs under gpurun_out/check_*).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 3000 python3 -m pytest tests -m gpu -x -q > gpurun_out/check_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/check_pytest.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/check_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/check_smoke.log
timeout 1500 python3 bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/check_bench.json
