#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=centroflye_amd/libcfhip.so; V=centroflye_amd/build_variants
timeout 600 python3 tools/dist_ab.py 50000 $V/r2.so $L $V/r2.so $L > gpurun_out/r3e_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/r3e_ab.log
rm -rf gpurun_out/pmc_cur
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d gpurun_out/pmc_cur -o p -- python3 tools/dist_ab.py 50000 $L > gpurun_out/r3e_pmc_cur.log 2>&1
python3 - <<'PY'
import csv, glob
acc = {}
for f in glob.glob("gpurun_out/pmc_cur/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cf_dist_kernel" in row["Kernel_Name"]:
            e = acc.setdefault(row["Counter_Name"], [0.0, set()])
            e[0] += float(row["Counter_Value"]); e[1].add(row["Dispatch_Id"])
print("cur", {k: (x[0] / max(1, len(x[1]))) for k, x in sorted(acc.items())})
PY
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q --durations=6 > gpurun_out/r3e_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r3e_pytest.log
