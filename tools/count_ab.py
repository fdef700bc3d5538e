#!/usr/bin/env python3
"""Developer tool: time A1 (cf_count_kmers) of several builds of the library on the same reads.
usage: tools/count_ab.py <reads> lib1.so lib2.so ...   (prints count ms of three calls, distinct k-mers, (read, k-mer) pairs)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
n = int(sys.argv[1])
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
for path in sys.argv[2:]:
    e = Engine(0, _lib.load(os.path.join(ROOT, path)))
    e.load(pk, 1)
    ms = []
    for _ in range(3):
        e.count_kmers(19)
        ms.append(round(e.times()["count_ms"], 2))
    st = e.stats()
    print(path, ms, st["n_distinct"], st["n_read_kmers"], flush=True)
    e.close()
