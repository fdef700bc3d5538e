#!/bin/bash
# Developer tool (GPU box): placement parity tests, then the bench's placement leg under a few knob settings.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -x -q -k "stage3 or placement or polisher or cli" > gpurun_out/place_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|error" gpurun_out/place_pytest.log | tail -5
for knobs in "${@:-place_mode=2}"; do
  args=""; for kv in $knobs; do args="$args --param $kv"; done
  timeout 600 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --place $args 2>gpurun_out/place_bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$knobs', d.get('placement'))"
done
