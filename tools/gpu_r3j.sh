#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $V/r2.so $L $L > gpurun_out/r3j_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3j_ab.log
python3 tools/dist_stamps.py 50000 2>&1 | tail -2
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --durations=5 > gpurun_out/r3j_pytest.log 2>&1; echo "pytest rc=$?"; tail -10 gpurun_out/r3j_pytest.log
timeout 900 python3 bench.py > gpurun_out/r3j_bench.json 2> gpurun_out/r3j_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3j_bench.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3j_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['stage_ms_per_step'], d['value_incl_transfers'], d['parity_vs_committed_oracle'], d.get('end_to_end'), d['cpu_baseline']['value'], d['cpu_baseline']['sample_matches_gpu'])
PY
