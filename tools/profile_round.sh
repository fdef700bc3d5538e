#!/usr/bin/env bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats of the default bench, the plain bench line, then PMC passes of
# ONE step (one rocprofv3 run per counter group, never combined with tracing domains: --kernel-trace only).
# Results land under gpurun_out/prof_<tag>/ ; tools/pmc_summary.py turns them into profiles/<tag>_*.
# usage: bash tools/profile_round.sh <tag>
set -u
tag=${1:-r06}
out=gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 bench.py --no-cpu-baseline --no-place --steps-b 0 --steps-c 0 > "$out/bench_under_rocprof.json" 2> "$out/stats.err"
# a first bench line: the PMC summary below takes the kernel times of this configuration from it; the round's line is taken again BEHIND the
# summary (bench.py copies the counters' derived figures of the latest profiles/rNN_pmc_*.json into its roofline blocks)
timeout 900 python3 bench.py --no-cpu-baseline --no-place > "$out/bench_line.json" 2> "$out/bench_first.err"
# ONE timed step of the bench's OWN configuration — every selected edge stored: the cap is the workload's edge count (3 156 871 804,
# profiles/r03_full_parity.json) + 1024, given explicitly so that the untimed sizing pass is skipped and every kernel is launched once
ONE="bench.py --steps 1 --warmup 0 --no-cpu-baseline --transfer-steps 0 --no-place --steps-b 0 --steps-c 0 --edge-cap 3156872828"
pmc() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$out/pmc_$name" -o "$name" -- python3 $ONE > /dev/null 2> "$out/pmc_$name.err"; }
pmc A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pmc B SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU TCC_HIT TCC_MISS
pmc C FETCH_SIZE
pmc D WRITE_SIZE
# the same four passes of cf_dist_kernel on the cenX-SHAPED reads of bench.py's workload_c (var_len 8; BASELINE configs[4]'s regime): ONE launch,
# every selected edge stored (1 809 974 541 of them, profiles/r06_parity_cenx_varlen8.json)
CENX="tools/cenx_probe.py --var-len 8 --once 1809975565"
pmcc() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$out/pmcc_$name" -o "$name" -- python3 $CENX > "$out/pmcc_$name.log" 2> "$out/pmcc_$name.err"; }
pmcc A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pmcc B SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU TCC_HIT TCC_MISS
pmcc C FETCH_SIZE
pmcc D WRITE_SIZE
python3 tools/pmc_summary.py "$tag" > "$out/pmc_summary_on_the_box.log" 2>&1
cp "$out/bench_line.json" "$out/bench_line_first.json"
timeout 900 python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err"
timeout 900 python3 bench.py --no-cpu-baseline --rr > "$out/bench_line_place_rr.json" 2> "$out/bench_place_rr.err"
find "$out" -name "*.csv" | head -40
