#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r2t_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r2t_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2t_bench.json 2> gpurun_out/r2t_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2t_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel_ms'], d['stage_ms_per_step'], d['counters']['n_edges'], d['counters']['n_unique'], d['counters']['n_dist_passes'])
PY
bash tools/gpu_cli_wall.sh 2>&1 | tail -25
