#!/usr/bin/env bash
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python3 tools/dist_ab.py 50000 $V/old.so $V/old_ss.so $V/old_sd.so $V/old_ds.so > gpurun_out/r2e_ab.log 2>&1
cat gpurun_out/r2e_ab.log
