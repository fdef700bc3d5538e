#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python3 tools/c2_stamps.py 50000 > gpurun_out/r2q.log 2>&1; cat gpurun_out/r2q.log
out=gpurun_out/prof_r2q; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --transfer-steps 0 --edge-cap 67108864 > $out/bench.json 2> $out/stats.err
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2q_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2q_pytest.log
