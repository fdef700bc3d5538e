#!/bin/bash
# per-kernel time table of one bench run (rocprofv3 --kernel-trace --stats), printed sorted by total time
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/stats_${1:-x}; rm -rf "$out"; mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 bench.py --no-cpu-baseline --no-place --transfer-steps 0 > "$out/bench.json" 2> "$out/err.log"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:32]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"])/1e6:9.3f} ms  total {float(r["TotalDurationNs"])/1e6:9.2f} ms  {float(r["Percentage"]):5.1f} %')
PY
