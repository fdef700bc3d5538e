#!/bin/bash
# Full GPU check: the whole -m gpu suite, then the bench line.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/full_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/full_pytest.log | tail -3
timeout 900 python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['whole_step_frac'], d['stage_ms_per_step'], d['value_incl_transfers']['value'], [l['value'] for l in d['cpu_baseline']['legs']])
PY
