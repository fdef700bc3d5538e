#!/usr/bin/env bash
# round 5, GPU call 1: new GPU tests (sharded stage script, checksums through the records), dist-kernel A/B, gview probe, copy modes
set -u
mkdir -p gpurun_out
V=centroflye_amd/build_variants
timeout 1500 python3 -m pytest tests/test_sharded_cli.py tests/test_exotic_symbols.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/c1_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/c1_tests.log
timeout 900 python3 tools/dist_ab.py 50000 $V/r04.so $V/new.so $V/nodrain2.so $V/nohotblk.so $V/nofillrd.so $V/pfa1.so $V/pfa3.so $V/pfb1.so $V/pfb3.so $V/pfa3b3.so $V/new.so $V/r04.so > gpurun_out/c1_dist_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/c1_dist_ab.log
timeout 900 python3 tools/gview_probe.py 500000 3 8 > gpurun_out/c1_gview.log 2>&1; echo "gview rc=$?"; grep case gpurun_out/c1_gview.log | cut -c1-260
for m in staged register; do CF_COPY_MODE=$m CF_COPY_THREADS=8 timeout 300 python3 tools/copy_bench.py x 2>&1 | tail -1 | sed "s/^/$m: /"; done | tee gpurun_out/c1_copy_modes.log
CF_COPY_MODE=staged CF_COPY_THREADS=16 timeout 300 python3 tools/copy_bench.py x 2>&1 | tail -1 | sed "s/^/staged16: /" | tee -a gpurun_out/c1_copy_modes.log
