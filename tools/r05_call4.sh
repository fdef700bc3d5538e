#!/usr/bin/env bash
set -u
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_n_motif2.py tests/test_dropin.py -m gpu -x -q -k "placement or n_motif or cli_scripts or stage_script" > gpurun_out/c4_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/c4_tests.log
timeout 600 python3 tools/place_bench.py 50000 place_mode=2 place_mode=2 > gpurun_out/c4_place_50k.log 2>&1; cat gpurun_out/c4_place_50k.log
timeout 900 python3 tools/place_bench.py 500000 place_mode=2 place_l3=2 > gpurun_out/c4_place_500k.log 2>&1; cat gpurun_out/c4_place_500k.log
timeout 900 python3 bench.py > gpurun_out/c4_bench_line.json 2> gpurun_out/c4_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/c4_bench_line.json
