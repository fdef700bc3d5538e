#!/bin/bash
# Developer tool (GPU box): tools/place_bench.py on the shipped library and on every variant in centroflye_amd/build_variants/pl2_*.so
cd "$(dirname "$0")/.."
n=${1:-50000}; shift
python tools/place_bench.py $n "${@:-place_mode=2}" 2>&1 | tail -8
for v in centroflye_amd/build_variants/pl2_*.so; do
  echo "== $v"; CF_DEBUG=1 CF_LIB=$v python tools/place_bench.py $n place_mode=2,place_spec=1 place_mode=1 2>&1 | grep -v "iter \|attempt\|stage " | tail -5
done
