#!/usr/bin/env python3
"""Developer tool: time the A1 count stage (and A3 clouds) of several builds of the library on the same reads.
usage: tools/count_ab.py <reads> lib1.so[:knob=value,...] lib2.so ...   (prints count ms / kernel ms, clouds ms, counters per library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
n = int(sys.argv[1])
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
for spec in sys.argv[2:]:
    path, _, knobs = spec.partition(":")
    e = Engine(0, _lib.load(os.path.join(ROOT, path)))
    for kv in filter(None, knobs.split(",")):
        e.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    e.load(pk, 1)
    rows = []
    for _ in range(3):
        e.count_kmers(19); t = e.times(); c = (round(t["count_ms"], 2), round(t["count_kernel_ms"], 2))
        n_rare = e.select_rare(3, 10, 32); e.build_clouds(); t = e.times()
        rows.append(c + (round(t["clouds_ms"], 2),))
    st = e.stats()
    print(spec, rows, st["n_distinct"], st["n_read_kmers"], n_rare, st["n_cloud_entries"], flush=True)
    e.close()
