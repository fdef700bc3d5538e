#!/usr/bin/env python3
"""Developer tool (GPU box): A3 (cf_build_clouds) on the bench's 50 000 reads per sparsity of the k-mer lookup table ("lut_shift": 16.0 / 13.8 / 13.3 / 13.0 ms at x 1 / 2 / 4 / 8;
a prefilter of 2 / 4 / 16 bits per slot instead of 8, measured with a knob that was not kept: 13.5 / 13.3 / 13.8); prints the stage's ms of three runs and the cloud checksum (identical in every setting)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from centroflye_amd import _host
from centroflye_amd.engine import Engine
pk = _host.synth(seed=2, n_units=15000, n_reads=50000, var_len=8)
for knobs in (dict(lut_shift=0), dict(lut_shift=1), dict(lut_shift=2), dict(lut_shift=3)):
    e = Engine(0)
    for k, v in knobs.items():
        e.set_param(k, v)
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32)
    ms = []
    for _ in range(3):
        n = e.build_clouds(); ms.append(round(e.times()["clouds_ms"], 2))
    print(knobs, ms, n, e.checksum("clouds"), flush=True)
    e.close()
