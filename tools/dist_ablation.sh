#!/usr/bin/env bash
# Runs ON THE GPU BOX: instruction counters of cf_dist_kernel for the shipped library and for the ablation builds
# (centroflye_amd/build_variants/abl<n>.so, tools/build_variant.sh abl<n> -DCF_DIST_ABL=<n>): one rocprofv3 --pmc pass each of
# tools/dist_ab.py on the bench's 50 000 reads.  tools/dist_ablation_summary.py turns the counters into profiles/r04_dist_phase_insts.md.
set -u
cd "$(dirname "$0")/.."
out=gpurun_out/abl; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
for v in full abl1 abl2 abl3 abl4 abl5 abl6 abl7; do
  lib=centroflye_amd/build_variants/$v.so; [ "$v" = full ] && lib=centroflye_amd/libcfhip.so
  [ -f "$lib" ] || continue
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d "$out/$v" -o p -- python3 tools/dist_ab.py 50000 $lib > "$out/$v.log" 2> "$out/$v.err"
  echo "$v rc=$? $(grep -v '^$' "$out/$v.log" | tail -1)"
done
find "$out" -name "*counter_collection.csv" | head -10
