#!/usr/bin/env bash
mkdir -p gpurun_out
python3 tools/c2_stamps.py 50000 centroflye_amd/build_variants/c2_exp1.so > gpurun_out/r2p.log 2>&1; cat gpurun_out/r2p.log
