#!/bin/bash
# Wall time of the read-recruitment CLI on the 1 Gb workload (BASELINE configs[1] shape): report parse / device / writers.
# Run on the GPU box:  gpurun -- 'bash tools/gpu_cli_wall.sh'
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/cli_wall; mkdir -p $OUT
W=/tmp/cfwall; mkdir -p $W
python - <<PY > $OUT/synth.log 2>&1
import time, os, sys
sys.path.insert(0, os.getcwd())
import bench
from centroflye_amd import _host
for n in (50000, 5000):
    t = time.time()
    _host.synth(report_path=f"$W/r{n}.ncrf", pack=False, n_reads=n, **bench.synth_kwargs(n, 1))
    print(n, "reads: report written in", round(time.time() - t, 2), "s,", os.path.getsize(f"$W/r{n}.ncrf"), "bytes", flush=True)
PY
cat $OUT/synth.log
for rep in 1 2; do
  T0=$(date +%s.%N); CFH_TIMING=1 python -m centroflye_amd.distance_based_kmer_recruitment --ncrf $W/r50000.ncrf --coverage 32 \
      --outdir $W/out50k --no-edges --metrics > $OUT/cli_noedges_$rep.log 2>&1
  echo "rc=$?"; cp $W/out50k/stage2_metrics.json $OUT/metrics_noedges_$rep.json
  python -c "import time,sys; print(\"wall\", round(time.time() - float(sys.argv[1]), 2), \"s\")" $T0 | tee $OUT/time_noedges_$rep.txt
done
# the binary pack cache (CF_PACK_CACHE): the first run writes it, the second loads it instead of parsing
for rep in 1 2; do
  CF_PACK_CACHE=$W/cache python -m centroflye_amd.distance_based_kmer_recruitment --ncrf $W/r50000.ncrf --coverage 32 --outdir $W/out50k --no-edges --metrics > $OUT/cli_cache_$rep.log 2>&1
  cp $W/out50k/stage2_metrics.json $OUT/metrics_cache_$rep.json
done
ls -l $W/cache
T0=$(date +%s.%N); CFH_TIMING=1 python -m centroflye_amd.distance_based_kmer_recruitment --ncrf $W/r5000.ncrf --coverage 32 \
      --outdir $W/out5k --metrics > $OUT/cli_edges.log 2>&1
echo "rc=$?"; cp $W/out5k/stage2_metrics.json $OUT/metrics_edges.json
python -c "import time,sys; print(\"wall\", round(time.time() - float(sys.argv[1]), 2), \"s\")" $T0 | tee $OUT/time_edges.txt
ls -l $W/out5k $W/out50k > $OUT/ls.txt; cat $OUT/ls.txt
grep cfhost $OUT/cli_noedges_2.log
python - <<PY
import json
for f in ("metrics_noedges_1", "metrics_noedges_2", "metrics_cache_1", "metrics_cache_2", "metrics_edges"):
    m = json.load(open(f"$OUT/{f}.json"))
    print(f, {k: m[k] for k in ("parse_s", "dist_wall_s", "total_s")}, m["times_ms"])
PY
