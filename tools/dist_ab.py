#!/usr/bin/env python3
"""Developer tool: time cf_dist_kernel of several builds of the library on the same reads.
usage: tools/dist_ab.py <reads> lib1.so[:knob=value,...] lib2.so ...   (prints kernel ms of two launches, the set-up ms, emissions, edges, passes, spilled per library)"""
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
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds()
    ms = []
    for _ in range(2):
        n_edges = e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
        ms.append(round(e.times()["dist_kernel_ms"], 1))
    ms.append("set-up %.1f ms" % e.times()["postings_ms"])      # postings + work lists of the last launch
    st = e.stats()
    print(spec, ms, st["n_emissions"], n_edges, st["n_dist_passes"], st["n_spilled"], flush=True)
    e.close()
