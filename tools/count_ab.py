#!/usr/bin/env python3
"""Developer tool (GPU box): A1 (cf_count_kmers) on the bench's 50 000 reads per library build; prints the stage's ms of three runs and the table checksum.
usage: tools/count_ab.py lib1.so lib2.so ...
(round 6: every workgroup of the hist / scatter kernels taking a CONTIGUOUS run of tiles instead of every gridDim-th one — so that the partial lines of
neighbouring tiles meet in one XCD's L2 —: 27.1-29.7 ms against 26.7-29.5: no difference, not kept)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
pk = _host.synth(seed=2, n_units=15000, n_reads=50000, var_len=8)
for path in sys.argv[1:] or ["centroflye_amd/libcfhip.so"]:
    e = Engine(0, _lib.load(os.path.join(ROOT, path)))
    e.load(pk, 1)
    ms = []
    for _ in range(3):
        e.count_kmers(19); ms.append(round(e.times()["count_ms"], 2))
    print(path, ms, e.checksum("table"), flush=True)
    e.close()
