#!/usr/bin/env bash
# round 5: the whole GPU suite and the bench line on the round's final tree
set -u
mkdir -p gpurun_out
timeout 2700 python3 -m pytest tests -m gpu -q --durations=12 > gpurun_out/final_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -18 gpurun_out/final_gpu_suite.log
timeout 900 python3 bench.py > gpurun_out/final_bench_line.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/final_bench_line.json
python3 -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
