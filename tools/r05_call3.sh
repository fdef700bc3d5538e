#!/usr/bin/env bash
# round 5, GPU call 3: the whole GPU suite (timed), placement phase stamps at 50 000 / 500 000 reads, launch shapes of the dist kernel for a rank of 8, the round's profiles
set -u
mkdir -p gpurun_out
timeout 2700 python3 -m pytest tests -m gpu -q --durations=30 > gpurun_out/c3_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -45 gpurun_out/c3_gpu_suite.log
CF_LIB=centroflye_amd/build_variants/pl2stamps.so timeout 600 python3 tools/place_bench.py 50000 place_mode=2 > gpurun_out/c3_place_stamps_50k.log 2>&1; echo "stamps50k rc=$?"; tail -6 gpurun_out/c3_place_stamps_50k.log | cut -c1-400
CF_LIB=centroflye_amd/build_variants/pl2stamps.so timeout 900 python3 tools/place_bench.py 500000 place_mode=2 > gpurun_out/c3_place_stamps_500k.log 2>&1; echo "stamps500k rc=$?"; tail -6 gpurun_out/c3_place_stamps_500k.log | cut -c1-400
timeout 900 python3 tools/gview_probe.py 500000 3 8 --shapes > gpurun_out/c3_shapes.log 2>&1; echo "shapes rc=$?"; grep case gpurun_out/c3_shapes.log | cut -c1-260
bash tools/profile_round.sh r05 > gpurun_out/c3_profile.log 2>&1; echo "profile rc=$?"; tail -5 gpurun_out/c3_profile.log
