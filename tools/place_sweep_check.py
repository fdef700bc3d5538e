#!/usr/bin/env python3
"""Developer tool (GPU box): the greedy placement against the C placer (oracle/c/cf_oracle_place.c) at a real size for parameter sets
OTHER than the defaults — min_cloud_kmer_freq, min_unit, min_inters (read_placer.py's --min-cloud-kmer-freq / --min-unit / --min-inters),
the multiplicity filter of the clouds, both device paths (place_mode 2 and 1).  Every line of read_positions.csv must be equal.
usage: tools/place_sweep_check.py [reads] [--units-per-read 0.3] [--out profiles/r04_place_sweep.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport
from conftest import lines_from_placement

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10000
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "place_sweep.json")
upr = float(sys.argv[sys.argv.index("--units-per-read") + 1]) if "--units-per-read" in sys.argv else 0.3      # array units per read: 0.3 = coverage 32; 1.5 = thin coverage, reads left unplaced
pk = _host.synth(seed=5, n_units=max(24, int(round(upr * n))), n_reads=n, var_len=8)
cls = pk.classify(50000)
rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
up, _, _, _ = pk.units(1)
SETS = [dict(min_mult=2, freq=2, min_unit=2, min_inters=10), dict(min_mult=2, freq=1, min_unit=1, min_inters=1), dict(min_mult=2, freq=3, min_unit=2, min_inters=30),
        dict(min_mult=2, freq=2, min_unit=4, min_inters=10), dict(min_mult=3, freq=2, min_unit=2, min_inters=80), dict(min_mult=1, freq=4, min_unit=3, min_inters=20)]
recs = []
with Engine(0) as e:
    e.load(pk, 1); e.count_kmers(19); e.select_rare(3, 10, 32); e.build_clouds(); e.reset_unique()
    e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, 0)
    gk = e.kmers()[e.unique_mask()]
    for s in SETS:
        e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(s["min_mult"])
        cp, ent = e.clouds()
        t0 = time.time()
        want = cport.place_reads(cls, rank, up, cp, ent, gk.size, s["freq"], s["min_unit"], s["min_inters"], 3)
        t_cpu = time.time() - t0
        wl = lines_from_placement(pk.ids, *[x.tolist() for x in want])
        r = dict(params=s, placed=sum(1 for x in wl if not x.endswith("None")), oracle_s=round(t_cpu, 1))
        for mode in (2, 1) + ((3,) if s["min_inters"] < 4 else ()):      # (3: the regions forced where the default routes to the hash-map path)
            e.set_param("place_mode", mode)
            try:
                got = e.place_reads(cls, rank, s["freq"], s["min_unit"], s["min_inters"], 3)
            except Exception as ex:      # (place_mode 3 may refuse: regions that would need more than 2^32 slots; 2 and 1 must not)
                if mode != 3:
                    raise
                r["mode3_refused"] = str(ex)[-120:]
                continue
            r[f"mode{mode}_identical"] = lines_from_placement(pk.ids, *[x.tolist() for x in got]) == wl
            r[f"mode{mode}_ms"] = round(float(e.times()["place_ms"]), 1)
        e.set_param("place_mode", 2)
        recs.append(r)
        print(json.dumps(r), flush=True)
ok = all(v for r in recs for kk, v in r.items() if kk.endswith("_identical"))
json.dump(dict(reads=n, array_units_per_read=upr, sets=recs, all_identical=ok), open(out, "w"), indent=1)
print("ALL IDENTICAL" if ok else "DIFFERENCES")
sys.exit(0 if ok else 1)
