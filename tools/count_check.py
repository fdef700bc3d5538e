#!/usr/bin/env python3
"""Developer tool (GPU box): A1 + A2 of <reads> synthetic reads on the GPU against the OpenMP oracle — counters, the checksum
over every (k-mer, pres, multi) of the table, the rare set element by element.  What tests/test_gpu_fullsize.py does at
50 000 reads, for sizes whose bucket count takes three sort passes (200 000 reads: 20 bits).  usage: tools/count_check.py <reads>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport
n = int(sys.argv[1])
P = dict(k=19, max_nonuniq=3, lo=10, hi=32)
pk = _host.synth(n_reads=n, seed=2, n_units=max(24, int(round(0.3 * n))), var_len=8)
up, us, ue, _ = pk.units(1)
e = Engine(0)
e.load(pk, 1)
t0 = time.time(); e.count_kmers(P["k"]); print("count ms", e.times()["count_ms"], flush=True)
keys, pres, multi = e.table(sort=False)
got = cport.table_checksum(keys, pres, multi); n_table = keys.size
del keys, pres, multi
n_rare = e.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
st = e.stats(); rare = e.kmers(); e.close()
t0 = time.time()
c, a = cport.stage2(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0, stop_after=1, want_arrays=True)
print("oracle s", round(time.time() - t0, 1), flush=True)
ok = ((st["n_bases"], st["n_windows"], st["n_read_kmers"]) == (c["n_bases"], c["n_windows"], c["n_read_kmers"]) and
      (st["n_distinct"], n_table, st["n_kept"], n_rare) == (c["n_distinct"], c["n_distinct"], c["n_kept"], c["n_rare"]) and
      got == c["table_checksum"] and np.array_equal(rare, a["rare"]))
print({"reads": n, "n_bases": st["n_bases"], "n_distinct": st["n_distinct"], "n_read_kmers": st["n_read_kmers"], "n_rare": n_rare, "table_checksum_equal": got == c["table_checksum"], "identical": bool(ok)})
sys.exit(0 if ok else 1)
