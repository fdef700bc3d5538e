#!/usr/bin/env python3
"""Developer tool: time cf_count_kmers / cf_select_rare / cf_build_clouds under several knob settings.
usage: tools/count_knobs.py <reads> "name=val,name=val" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host
from centroflye_amd.engine import Engine
n = int(sys.argv[1])
pk = _host.synth(seed=2, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
e = Engine(0)
e.load(pk, 1)
for setting in sys.argv[2:]:
    for k, v in [x.split("=") for x in setting.split(",") if x]:
        e.set_param(k, int(v))
    for _ in range(2):
        e.count_kmers(19); t = e.times(); c = t["count_ms"]
        nr = e.select_rare(3, 10, 32); s = e.times()["select_ms"]
        ne = e.build_clouds(); cl = e.times()["clouds_ms"]
    print(setting, "count", round(c, 1), "select", round(s, 1), "clouds", round(cl, 1), nr, ne, flush=True)
