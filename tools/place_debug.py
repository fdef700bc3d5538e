#!/usr/bin/env python3
"""Developer tool (GPU box): placement of N synthetic reads by several builds / knob sets of the library and by the C
oracle; prints where they differ.  usage: tools/place_debug.py <reads> <seed> lib[:knob=v,...] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
from oracle import cport
from conftest import lines_from_placement
n, seed = int(sys.argv[1]), int(sys.argv[2])
pk = _host.synth(seed=seed, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
up, _, _, _ = pk.units(1)
cls = pk.classify(50000)
rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
outs = {}
cp = ent = gk = None
for spec in sys.argv[3:]:
    path, _, knobs = spec.partition(":")
    e = Engine(0, _lib.load(os.path.join(ROOT, path)))
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
    e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
    gk = e.kmers()[e.unique_mask()]
    e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
    cp, ent = e.clouds()
    for kv in filter(None, knobs.split(",")):
        e.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    for rep in range(2):
        t0 = time.time()
        got = e.place_reads(cls, rank, 2, 2, 10, 3)
        outs[f"{spec}#{rep}"] = lines_from_placement(pk.ids, *[x.tolist() for x in got])
        print(spec, rep, f"{time.time() - t0:.2f} s", flush=True)
    e.close()
t0 = time.time()
want = cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)
outs["oracle"] = lines_from_placement(pk.ids, *[x.tolist() for x in want])
print("oracle", f"{time.time() - t0:.1f} s", flush=True)
names = list(outs)
for a in names:
    for b in names:
        if a >= b: continue
        la, lb = outs[a], outs[b]
        d = [i for i, (x, y) in enumerate(zip(la, lb)) if x != y]
        print(f"{a} vs {b}: {len(d)} differing lines" + (f", first at {d[0]}: {la[d[0]]!r} / {lb[d[0]]!r}" if d else ""), flush=True)
