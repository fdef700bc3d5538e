#!/usr/bin/env bash
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python3 tools/dist_ab.py 50000 $V/old.so $V/new_sd.so $V/new_ss.so $V/new_dd.so $V/new_sd_stamps.so > gpurun_out/r2d_ab.log 2>&1
cat gpurun_out/r2d_ab.log
