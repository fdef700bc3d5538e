#!/usr/bin/env python3
"""Developer tool: time cf_dist_kernel under several knob settings on the same reads.
usage: tools/dist_knobs.py <reads> "name=val,name=val" ...   (one line per setting: kernel ms, edges, passes, spilled)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
n = int(sys.argv[1])
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
e = Engine(0, _lib.load(os.path.join(ROOT, os.environ['CF_LIB']))) if os.environ.get('CF_LIB') else Engine(0)
e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds()
for setting in sys.argv[2:]:
    kv = [x.split("=") for x in setting.split(",") if x]
    for k, v in kv:
        e.set_param(k, int(v))
    ms = []
    for _ in range(2):
        n_edges = e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
        ms.append(round(e.times()["dist_kernel_ms"], 1))
    st = e.stats()
    print(setting, ms, st["n_emissions"], n_edges, st["n_dist_passes"], st["n_spilled"], flush=True)
