#!/usr/bin/env bash
# round 5, GPU call 2: dist-kernel A/B (second batch), placement with the third level, the var_len 1 oracle record, the bench line, the whole GPU suite
set -u
mkdir -p gpurun_out
V=centroflye_amd/build_variants
timeout 900 python3 tools/dist_ab.py 50000 $V/base.so $V/pfa1.so $V/pfb1.so $V/pfa1b1.so $V/nofillrd.so $V/pushflat.so $V/pushflatb1.so $V/pb8.so $V/base.so > gpurun_out/c2_dist_ab.log 2>&1; echo "ab rc=$?"; cat gpurun_out/c2_dist_ab.log
timeout 600 python3 tools/place_bench.py 50000 place_mode=2 place_l3=1 place_l3=1,place_l3_shift=3 > gpurun_out/c2_place_50k.log 2>&1; echo "place50k rc=$?"; cat gpurun_out/c2_place_50k.log
timeout 900 python3 tools/place_bench.py 500000 place_mode=2 place_l3=2 place_l3=0 > gpurun_out/c2_place_500k.log 2>&1; echo "place500k rc=$?"; cat gpurun_out/c2_place_500k.log
timeout 1200 python3 tools/parity_record.py --reads 50000 --seed 2 --var-len 1 --part 0 --n-parts 1 --sub 8 --gpu --out gpurun_out/r05_parity_50k_varlen1.json > gpurun_out/c2_record_varlen1.log 2>&1; echo "record rc=$?"; tail -2 gpurun_out/c2_record_varlen1.log | cut -c1-1200
timeout 900 python3 bench.py > gpurun_out/c2_bench_line.json 2> gpurun_out/c2_bench.err; echo "bench rc=$?"; cut -c1-3000 gpurun_out/c2_bench_line.json; tail -3 gpurun_out/c2_bench.err
timeout 2400 python3 -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/c2_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -40 gpurun_out/c2_gpu_suite.log
