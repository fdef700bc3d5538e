#!/usr/bin/env python3
"""Developer tool (GPU box): the greedy placement on THIN coverage — 15 000 reads of mean 40 kb over a 45 000-unit array (coverage 6.5): stage 2's
"unique" k-mers are not unique to one place of the array there, a read meets the contig at many offsets and has dozens of candidate score rows
(round 5, tools/fuzz_place.py: 4.1 s on the region path against 0.58 s on the hash-map path, to which such runs were then handed).  Round 6: the
tail gives such reads a WAVE each (a lane per row).  Prints device ms per place_mode (2: regions with the hand-over rule, 3: regions always,
1: hash map) and checks every line against the C placer.
usage: tools/place_thin.py [reads=15000] [--freq 2] [--prefix-threshold 5000] [--var-len 8] [--units-per-read 3]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport
from conftest import lines_from_placement
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 15000
def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
freq, pthr, vlen, upr = opt("--freq", 2), opt("--prefix-threshold", 5000), opt("--var-len", 8), opt("--units-per-read", 3)
pk = _host.synth(seed=11, n_reads=n, n_units=upr * n, mean_len=40000.0, var_len=vlen)
cls = pk.classify(pthr)
rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
up = pk.units(1)[0]
with Engine(0) as e:
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
    e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
    gk = e.kmers()[e.unique_mask()]
    e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
    cp, ent = e.clouds()
    t0 = time.time()
    want = lines_from_placement(pk.ids, *[x.tolist() for x in cport.place_reads(cls, rank, up, cp, ent, gk.size, freq, 2, 10, 3)])
    print(json.dumps(dict(reads=int(pk.n_reads), bases=int(pk.n_bases), unique_kmers=int(gk.size), cloud_entries=int(ent.size), placed=sum(1 for x in want if not x.endswith("None")),
                          c_placer_s=round(time.time() - t0, 2))), flush=True)
    for mode in (2, 3, 1, 2):
        e.set_param("place_mode", mode)
        t0 = time.time()
        got = lines_from_placement(pk.ids, *[x.tolist() for x in e.place_reads(cls, rank, freq, 2, 10, 3)])
        print(json.dumps(dict(place_mode=mode, device_ms=round(float(e.times()["place_ms"]), 1), wall_s=round(time.time() - t0, 3), identical=got == want)), flush=True)
    e.set_param("place_mode", 2)
