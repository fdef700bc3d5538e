#!/bin/bash
# Larger single-GPU workloads (same generator, coverage 32): 200 000 and 400 000 reads per GPU.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for n in 200000 400000; do
  timeout 900 python bench.py --reads $n --steps 2 --warmup 1 --no-cpu-baseline --transfer-steps 0 --no-place --edge-cap 67108864 > gpurun_out/big_$n.json 2> gpurun_out/big_$n.err
  echo "rc=$? reads=$n"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/big_$n.json").read().strip().splitlines()[-1])
    print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["stage_ms_per_step"], {k:d["counters"][k] for k in ("n_bases","n_distinct","n_rare","n_cloud_entries","n_emissions","n_edges","n_dist_passes")}, d["steps_identical"])
except Exception as e:
    print("no line:", e); print(open("gpurun_out/big_$n.err").read()[-800:])
PY
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
