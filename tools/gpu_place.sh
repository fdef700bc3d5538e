#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -x -q -k "stage3 or placement or polisher or cli" > gpurun_out/place_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/place_pytest.log | tail -2
for f in 1 0; do
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --place --param place_fused=$f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f', d.get('placement'))"
done
