#!/usr/bin/env python3
"""Developer tool (GPU box): host <-> device copy rates of the C ABI's staged copies (cf_load_reads / cf_get_edges) for several
thread counts, into fresh and into already-touched numpy buffers."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from centroflye_amd import _host
    from centroflye_amd.engine import Engine
    pk = _host.synth(seed=2, n_units=6000, n_reads=20000, var_len=8)
    e = Engine(0)
    t0 = time.perf_counter(); e.load(pk, 1); t1 = time.perf_counter()
    t2 = time.perf_counter(); e.load(pk, 1); t3 = time.perf_counter()
    e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds()
    ne = e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=300_000_000)
    n = min(ne, 250_000_000)
    t4 = time.perf_counter(); a = e.edges(n); t5 = time.perf_counter()
    out = np.empty((n, 4), np.uint32); out[:] = 0
    t6 = time.perf_counter(); e._check(e._lib.cf_get_edges(e._ctx, out.ctypes.data, n), "get"); t7 = time.perf_counter()
    print(f"threads {os.environ.get('CF_COPY_THREADS', '8'):>2}: H2D {pk.n_bases / 1e9:.2f} GB first {pk.n_bases / (t1 - t0) / 1e9:5.1f} GB/s again {pk.n_bases / (t3 - t2) / 1e9:5.1f} GB/s | "
          f"D2H {n * 16 / 1e9:.2f} GB fresh buffer {n * 16 / (t5 - t4) / 1e9:5.1f} GB/s touched buffer {n * 16 / (t7 - t6) / 1e9:5.1f} GB/s", flush=True)
else:
    for t in (1, 4, 8, 12, 16):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, CF_COPY_THREADS=str(t)))
