#!/usr/bin/env python3
"""Developer tool (GPU box): A3 clouds and ONE first-k-mer partition of the distance stage (A5/A6) of <reads> synthetic reads on
the GPU against the OpenMP oracle's same partition — what tests/test_gpu_fullsize.py does at 50 000 reads, for sizes whose
k-mer set needs the other table layouts (200 000 reads: 2.2e7 rare k-mers, 6-byte slots with a 7-bit distance field).
usage: tools/dist_part_check.py <reads> <part> <n_parts>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport
n, part, n_parts = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
pk = _host.synth(n_reads=n, seed=2, n_units=max(24, int(round(0.3 * n))), var_len=8)
up, us, ue, _ = pk.units(1)
e = Engine(0)
e.load(pk, 1); e.count_kmers(P["k"]); n_rare = e.select_rare(P["max_nonuniq"], P["lo"], P["hi"]); n_ce = e.build_clouds()
t0 = time.time()
with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0) as st:
    print("oracle A1-A3 s", round(time.time() - t0, 1), flush=True)
    c = st.counters; a = st.arrays()
    ok = (n_rare, n_ce) == (c["n_rare"], c["n_cloud_entries"]) and np.array_equal(e.kmers(), a["rare"])
    cp, ent = e.clouds()
    ok = ok and np.array_equal(cp, a["cloud_ptr"]) and np.array_equal(ent, a["entries"])
    del cp, ent, a
    print("clouds identical", bool(ok), flush=True)
    uq = np.zeros(n_rare, np.uint8)
    t0 = time.time()
    w = st.dist_part(part, n_parts, 0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], threads=0, unique=uq)
    print("oracle partition s", round(time.time() - t0, 1), flush=True)
e.reset_unique()
ne = e.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], part, n_parts, edge_cap=w["n_edges"] + 16)
s2 = e.stats()
ok = ok and (ne, s2["n_emissions"], s2["n_unique"]) == (w["n_edges"], w["n_emissions"], w["n_unique"]) and e.edges_checksum() == w["edge_checksum"]
ok = ok and np.array_equal(e.unique_mask(), uq.astype(bool))
print({"reads": n, "part": part, "n_parts": n_parts, "n_rare": n_rare, "n_cloud_entries": n_ce, "n_emissions": s2["n_emissions"], "n_edges": ne, "n_unique": s2["n_unique"],
       "dist_kernel_ms": e.times()["dist_kernel_ms"], "identical": bool(ok)})
e.close()
sys.exit(0 if ok else 1)
