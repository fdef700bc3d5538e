#!/usr/bin/env bash
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python3 tools/dist_ab.py 50000 $V/old_ss.so $V/cur.so $V/cur_stamps.so > gpurun_out/r2f_ab.log 2>&1
cat gpurun_out/r2f_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_dropin.py -m gpu -x -q -k "not config1" 2>&1 | tail -3
