#!/usr/bin/env python3
"""Diagnostic: run stage 2 with the -DCF_DIST_STAMPS build (centroflye_amd/libcfhip_diag.so) and print the
per-phase shader-clock shares of cf_dist_kernel (stderr line '[cf_dist stamps] ...').  Never a timing source."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
e = Engine(0, _lib.load(os.path.join(ROOT, "centroflye_amd", "libcfhip_diag.so")))
e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds()
e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
print(e.stats()["n_emissions"], e.times()["dist_kernel_ms"])
