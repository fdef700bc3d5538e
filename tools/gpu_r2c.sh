#!/usr/bin/env bash
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python3 tools/dist_ab.py 50000 $V/old_stamps.so $V/new_stamps.so $V/new_bd_stamps.so $V/new_stamps.so:dist_wgs=1,dist_block=1024 $V/old.so $V/new.so > gpurun_out/r2c_ab.log 2>&1
cat gpurun_out/r2c_ab.log
