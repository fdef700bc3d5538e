#!/usr/bin/env bash
# usage: tools/sweep.sh <reads> <param> v1 v2 ...   (developer tool: one bench line per knob value)
reads=$1; param=$2; shift 2
for v in "$@"; do
  timeout 600 python bench.py --reads $reads --steps 1 --warmup 1 --no-cpu-baseline --param $param=$v 2>&1 | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], round(d['roofline']['kernel_ms'],2), '%.3g'%d['roofline']['pair_emissions_per_s'], d['counters']['n_dist_passes'], d['counters']['n_spilled'])" $param $v
done
