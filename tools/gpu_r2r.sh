#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py tests/test_exotic_symbols.py -m gpu -x -q -k "stage3 or placement or cli or exotic or max_distance" > gpurun_out/r2r_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2r_pytest.log
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --transfer-steps 0 --edge-cap 1000 --place > gpurun_out/r2r_place1.json 2> gpurun_out/r2r_place1.err; echo rc=$?
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --transfer-steps 0 --edge-cap 1000 --place --param place_mode=0 > gpurun_out/r2r_place0.json 2> gpurun_out/r2r_place0.err; echo rc=$?
python3 -c "
import json
for f in ('gpurun_out/r2r_place1.json','gpurun_out/r2r_place0.json'):
    d=json.load(open(f)); print(f, d['placement'])"
