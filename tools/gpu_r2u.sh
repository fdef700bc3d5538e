#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/dist_ab.py 50000 centroflye_amd/build_variants/diag_base.so centroflye_amd/build_variants/diag_sk1.so centroflye_amd/build_variants/diag_sk2.so centroflye_amd/build_variants/diag_tb1.so centroflye_amd/build_variants/diag_sk1tb1.so > gpurun_out/r2u_ab.log 2>&1
cat gpurun_out/r2u_ab.log | grep -v "^$" | tail -30
echo "--- disk"
( time dd if=/dev/zero of=/tmp/ddtest bs=1M count=6000 conv=fsync ) 2>&1 | tail -5; rm -f /tmp/ddtest
( time dd if=/dev/zero of=/dev/shm/ddtest bs=1M count=6000 ) 2>&1 | tail -5; rm -f /dev/shm/ddtest
df -h /tmp /dev/shm | tail -3
W=/dev/shm/cfwall; mkdir -p $W
python - <<PY
import time, os, sys
sys.path.insert(0, os.getcwd())
import bench
from centroflye_amd import _host
_host.synth(report_path="$W/r5000.ncrf", pack=False, n_reads=5000, **bench.synth_kwargs(5000, 1))
PY
T0=$(date +%s.%N); python -m centroflye_amd.distance_based_kmer_recruitment --ncrf $W/r5000.ncrf --coverage 32 --outdir $W/out5k --metrics > gpurun_out/r2u_cli.log 2>&1
python -c "import time,sys; print(\"wall shm\", round(time.time() - float(sys.argv[1]), 2), \"s\")" $T0
ls -l $W/out5k; cat $W/out5k/stage2_metrics.json | head -5; rm -rf $W
