#!/usr/bin/env python3
"""Developer tool (GPU box): stage 2 + A3/A4 once, then the greedy placement once per knob setting given on the command
line ("place_mode=2,place_grid=16" ...); prints device ms and us per placed read, and checks that every setting gives the
same lines.  usage: tools/place_bench.py <reads> [knobs ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine

n = int(sys.argv[1])
settings = sys.argv[2:] or ["place_mode=2"]
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
cls = pk.classify(50000)
rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
lib = None
if os.environ.get("CF_LIB"):
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
e = Engine(0, lib)
e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
gk = e.kmers()[e.unique_mask()]
e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
first = None
for s in settings:
    for kv in s.split(","):
        k, v = kv.split("="); e.set_param(k, int(v))
    t0 = time.time()
    got = e.place_reads(cls, rank, 2, 2, 10, 3)
    dt = time.time() - t0
    ms = e.times()["place_ms"]
    same = True if first is None else all(np.array_equal(a, b) for a, b in zip(first, got))
    if first is None:
        first = got
    print(f"{s:50s} device {ms:9.1f} ms  wall {dt:7.3f} s  {1e3 * ms / max(1, n):7.2f} us per read  same_as_first={same}", flush=True)
e.close()
