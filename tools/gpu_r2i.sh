#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2l_pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r2l_pytest.log | tail -3
out=gpurun_out/prof_r2l; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 --transfer-steps 0 --edge-cap 67108864 > $out/bench.json 2> $out/stats.err
python3 -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['steps_identical'])"
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-70,150-260
