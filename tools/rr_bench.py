#!/usr/bin/env python3
"""Developer tool: throughput of cf_rr_distances (read recruitment) on synthetic HOR reads plus as
many random reads, against the C oracle (and the reference's edlib when oracle/_ref is present) on a small sample.
usage: tools/rr_bench.py <reads>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import rr
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
sample = list(range(0, n, max(1, n // 20)))[:20] + [n + i for i in range(5)]
t = time.time(); want = [(rr.distance(unit, flat[off[i]:off[i + 1]].tobytes(), 350), rr.distance(rr.revcomp(unit), flat[off[i]:off[i + 1]].tobytes(), 350)) for i in sample]; dt = time.time() - t
sb = sum(int(off[i + 1] - off[i]) for i in sample)
print("sample equal to the C oracle:", want == [(int(fwd[i]), int(rc[i])) for i in sample], f"; oracle {sb / dt / 1e6:.2f} Mbases/s on one core")
if rr.ref_distance(b"ACGT", b"ACGT", 1) is not None:
    t = time.time(); ref = [(rr.ref_distance(unit, flat[off[i]:off[i + 1]].tobytes(), 350), rr.ref_distance(rr.revcomp(unit), flat[off[i]:off[i + 1]].tobytes(), 350)) for i in sample]; dt = time.time() - t
    print("sample equal to the reference's edlib:", ref == want, f"; edlib {sb / dt / 1e6:.2f} Mbases/s on one core")
