#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=centroflye_amd/build_variants
python tools/dist_ab.py 50000 $V/dp_a1b1.so $V/dp_a2b1.so $V/dp_a4b1.so $V/dp_a6b1.so $V/dp_a8b1.so $V/dp_a4b2.so > gpurun_out/r2v_ab.log 2>&1
grep -v "^$" gpurun_out/r2v_ab.log | tail -20
