import sys, os
sys.path.insert(0, "/root/repo")
from centroflye_amd import _host
from centroflye_amd.engine import Engine
pk = _host.synth(seed=2, n_units=15000, n_reads=50000, var_len=8)
for sh in (0, 1, 2):
    e = Engine(0)
    e.set_param("lut_shift", sh)
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32)
    ms = []
    for _ in range(3):
        n = e.build_clouds(); ms.append(round(e.times()["clouds_ms"], 2))
    print("lut_shift", sh, ms, n, e.checksum("clouds"))
    e.close()
