#!/bin/bash
# End-to-end check on the GPU box: stage 2 and stage 3 CLIs twice on a 5 000-read synthetic report; the outputs of the two runs must be identical.
set -e
cd /root/repo
W=/tmp/e2e; rm -rf $W; mkdir -p $W
python - <<PY
import sys, os
sys.path.insert(0, os.getcwd())
import bench
from centroflye_amd import _host
_host.synth(report_path="$W/r.ncrf", pack=False, n_reads=5000, **bench.synth_kwargs(5000, 3))
PY
for i in 1 2; do
  python scripts/distance_based_kmer_recruitment.py --ncrf $W/r.ncrf --coverage 32 --outdir $W/s2_$i --no-edges > $W/s2_$i.log 2>&1
  python scripts/read_placer.py --ncrf $W/r.ncrf --genomic-kmers $W/s2_$i/unique_kmers_min_edge_cov_4.txt --outdir $W/s3_$i > $W/s3_$i.log 2>&1
done
cmp $W/s2_1/unique_kmers_min_edge_cov_4.txt $W/s2_2/unique_kmers_min_edge_cov_4.txt && echo "kmers identical"
cmp $W/s3_1/read_positions.csv $W/s3_2/read_positions.csv && echo "placements identical"
wc -l $W/s2_1/unique_kmers_min_edge_cov_4.txt $W/s3_1/read_positions.csv
grep -c None $W/s3_1/read_positions.csv || true
head -3 $W/s3_1/read_positions.csv
