#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r2h_bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['steps_identical'], d['counters'])"
out=gpurun_out/prof_r2h; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 0 --transfer-steps 0 > /dev/null 2> $out/stats.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); head -25 "$f" | cut -c1-160
