#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in "place_chunk=2" "place_chunk=1" "place_chunk=2 --param place_grid=256" "place_chunk=1 --param place_grid=256" "place_chunk=2 --param place_grid=512" "place_chunk=1 --param place_grid=512" "place_chunk=2" "place_chunk=4 --param place_grid=512" "place_fused=0"; do
  python bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --place --param $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', round(d['placement']['place_device_ms'],1), d['placement']['placed'])"
done
