#!/usr/bin/env python3
"""Diagnostic: A1 by sort and reduce with the -DCF_C2_STAMPS build: per-phase shader-clock shares of cf_c2_reduce_kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
e = Engine(0, _lib.load(os.path.join(ROOT, sys.argv[2] if len(sys.argv) > 2 else "centroflye_amd/build_variants/c2_stamps.so")))
e.load(pk, 1)
for _ in range(2):
    e.count_kmers(19)
    print(e.times()["count_ms"], e.times()["count_kernel_ms"], e.stats()["n_read_kmers"])
