#!/usr/bin/env bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r2s_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r2s_pytest.log
bash tools/profile_round.sh r02 > gpurun_out/r2s_profile.log 2>&1
tail -3 gpurun_out/prof_r02/bench_line.json | cut -c1-400
