#!/usr/bin/env python3
"""Developer tool (GPU box): the shape of the placement workload.  Runs stage 2 + A3/A4 + the greedy placement on the
device for <reads> synthetic reads, then replays the placement order on the host (tools/place_stats.c) and writes
gpurun_out/place_stats_<reads>.json: entries / events / postings / hits per greedy iteration, postings per k-mer, score
rows per read and how far their offsets lie from the read's final offset.  usage: tools/place_stats.py <reads> [seed]"""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine

n = int(sys.argv[1]); seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
so = os.path.join(ROOT, "tools", "place_stats.so")
subprocess.check_call(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "place_stats.c")])
L = C.CDLL(so)
pk = _host.synth(seed=seed, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
up = np.ascontiguousarray(np.asarray(pk.units(1)[0]), np.int64)
cls = np.ascontiguousarray(pk.classify(50000), np.uint8)
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
cp, ent = e.clouds()
cp = np.ascontiguousarray(cp, np.int64); ent = np.ascontiguousarray(ent, np.int32)
t0 = time.time()
rd, pos, s0, s1 = e.place_reads(cls, rank, 2, 2, 10, 3)
t_place = time.time() - t0
place_ms = e.times().get("place_ms")
e.close()
rd = np.ascontiguousarray(rd, np.int64); pos = np.ascontiguousarray(pos, np.int64)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", f"place_stats_{n}.json")
P = C.c_void_p
L.place_stats.argtypes = [C.c_int64, C.c_int64, P, P, P, P, P, P, C.c_int, C.c_char_p]
rc = L.place_stats(n, int(gk.size), cls.ctypes.data, up.ctypes.data, cp.ctypes.data, ent.ctypes.data, rd.ctypes.data, pos.ctypes.data, 2, out.encode())
assert rc == 0
d = json.load(open(out))
d["place_s"] = t_place; d["place_device_ms"] = place_ms; d["classes"] = np.bincount(cls, minlength=3).tolist()
json.dump(d, open(out, "w"), indent=1)
print(json.dumps(d))
