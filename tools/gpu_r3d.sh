#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
CF_PLACE_DEBUG=13447,35727,31523 timeout 900 python3 tools/place_check.py 50000 2 > gpurun_out/r3d_place_dbg.log 2>&1; echo "place dbg rc=$?"; grep -v "^$" gpurun_out/r3d_place_dbg.log | head -60
for v in r2 cur; do
  lib=centroflye_amd/build_variants/r2.so; [ $v = cur ] && lib=centroflye_amd/libcfhip.so
  rm -rf gpurun_out/pmc_$v
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD -d gpurun_out/pmc_$v -o p -- python3 tools/dist_ab.py 50000 $lib > gpurun_out/r3d_pmc_$v.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
v = sys.argv[1]
acc = {}
for f in glob.glob(f"gpurun_out/pmc_{v}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "cf_dist_kernel" in row["Kernel_Name"]:
            e = acc.setdefault(row["Counter_Name"], [0.0, set()])
            e[0] += float(row["Counter_Value"]); e[1].add(row["Dispatch_Id"])
print(v, {k: (x[0] / max(1, len(x[1]))) for k, x in sorted(acc.items())})
PY
done
