#!/usr/bin/env bash
# round 2, GPU call B: new dist kernel (byte counters on hash(b), bitmap pass, static items): parity, timing, instruction counts
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-r2b}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not config1" > gpurun_out/${tag}_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo "bench rc=$?" >> gpurun_out/${tag}_bench.err
out=gpurun_out/prof_${tag}; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU TCC_HIT TCC_MISS -d $out/pmc_B -o B -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --edge-cap 67108864 --transfer-steps 0 > /dev/null 2> $out/pmc_B.err
python3 - <<PY
import csv, glob
acc = {}
for f in glob.glob("$out/pmc_B/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:40], row["Counter_Name"])
        acc[k] = acc.get(k, 0.0) + float(row["Counter_Value"])
for k in sorted(acc):
    if k[1] in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE"): print(k, "%.4g" % acc[k])
PY
tail -3 gpurun_out/${tag}_pytest.log; python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_bench.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['roofline']['kernel_ms'], d['steps_identical'], d['value_incl_transfers'])"
