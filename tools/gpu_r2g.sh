#!/usr/bin/env bash
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python3 tools/dist_ab.py 50000 $V/old_ss.so $V/v_00.so $V/v_02.so $V/v_10.so $V/v_12.so $V/v_11.so > gpurun_out/r2g_ab.log 2>&1
cat gpurun_out/r2g_ab.log
