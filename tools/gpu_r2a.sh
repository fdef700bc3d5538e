#!/usr/bin/env bash
# round 2, GPU call A: host facts, the whole -m gpu suite (incl. the full-size parity tests), one default bench line
mkdir -p gpurun_out
{ nproc; free -g | head -2; python -c "import os;print(len(os.sched_getaffinity(0)))"; } > gpurun_out/r2a_host.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2a_pytest.log
timeout 600 python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
echo "bench rc=$?" >> gpurun_out/r2a_bench.err
tail -5 gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_host.txt; head -c 600 gpurun_out/r2a_bench.json
