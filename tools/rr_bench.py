#!/usr/bin/env python3
"""Developer tool: throughput of cf_rr_distances (read recruitment) on synthetic HOR reads plus as
many random reads (`bench.py --rr` reports the same with the CPU baseline beside it).
usage: tools/rr_bench.py <reads>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
unit = pk.motifs[0].encode()
rng = np.random.default_rng(3)
rand = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, pk.n_bases)]
flat = np.concatenate([pk.bases, rand])
off = np.concatenate([pk.read_off, pk.read_off[1:] + pk.n_bases])
e = Engine(0)
for _ in range(2):
    t = time.time(); fwd, rc = e.rr_distances(unit, flat, off, 350); dt = time.time() - t
    ms = e.times()["rr_kernel_ms"]
    print(f"{off.size - 1} reads, {flat.size / 1e6:.0f} Mbases (both strands each): kernel {ms:.1f} ms = {flat.size / ms / 1e6:.2f} Gbases/s; call {dt:.2f} s; "
          f"recruited {int(((fwd != -1) | (rc != -1)).sum())} (HOR reads {n})", flush=True)
