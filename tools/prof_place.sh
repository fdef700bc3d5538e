cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_place -o pl -- python3 tools/place_bench.py 50000 place_mode=2 > gpurun_out/prof_place.log 2>&1
tail -3 gpurun_out/prof_place.log
find gpurun_out/prof_place -name "*kernel_stats*" | head
f=$(find gpurun_out/prof_place -name "*kernel_stats.csv" | head -1); grep -i "pl2\|Name" $f | head -20
