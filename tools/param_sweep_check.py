#!/usr/bin/env python3
"""Developer tool (GPU box): the stage-2 path against the OpenMP oracle at a real size for parameter sets OTHER than the benchmark's —
other k (the 2-bit code's limits, other table layouts), other rare windows, min_cov, max_d, a threshold that takes the double
division instead of 5 cnt >= 4 total.  Per set: tests/bigparity.check (A1 table checksum, A2 rare set, A3 CSR, one first-k-mer partition of
A5/A6: counters, edge checksum, unique bits).  usage: tools/param_sweep_check.py [reads] [--synth key=value,...] [--sets i,j,...] [--out profiles/r04_param_sweep.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from centroflye_amd import _host
from centroflye_amd.engine import Engine
import bigparity

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", "param_sweep.json")
BASE = dict(bigparity.P)
SETS = [dict(k=15), dict(k=25), dict(k=31), dict(min_cov=2, max_d=40), dict(rel_threshold=0.6), dict(max_nonuniq=0, lo=6, hi=20),
        dict(k=11, lo=20, hi=200, max_d=20), dict(min_d=3, max_d=300, min_cov=6)]
sy = dict(seed=11, n_units=max(24, int(round(0.3 * n))), n_reads=n, var_len=8)
if "--synth" in sys.argv:      # other read sets: --synth mean_len=80000,max_len=400000,p_sub=0.05 ...
    for kv in sys.argv[sys.argv.index("--synth") + 1].split(","):
        kk, vv = kv.split("=")
        sy[kk] = float(vv) if "." in vv else int(vv)
if "--sets" in sys.argv:       # indices into SETS (and -1: the benchmark's own parameters)
    pick = [int(x) for x in sys.argv[sys.argv.index("--sets") + 1].split(",")]
    SETS = [dict() if i < 0 else SETS[i] for i in pick]
pk = _host.synth(**sy)
recs = []
with Engine(0) as e:
    e.load(pk, 1)
    for i, s in enumerate(SETS):
        bigparity.P.clear(); bigparity.P.update(BASE); bigparity.P.update(s)
        t0 = time.time()
        r = bigparity.check(e, pk, part=5, n_parts=16, loaded=True)
        r["params"] = dict(bigparity.P); r["seconds"] = round(time.time() - t0, 1)
        recs.append(r)
        print(json.dumps(dict(params=s, identical=r["identical"], n_rare=r["n_rare"], n_emissions_partition=r["n_emissions_partition"],
                              n_edges_partition=r["n_edges_partition"], checks=r["checks"], s=r["seconds"])), flush=True)
json.dump(dict(reads=n, synth={k: v for k, v in sy.items()}, sets=recs, all_identical=all(r["identical"] for r in recs)), open(out, "w"), indent=1)
print("ALL IDENTICAL" if all(r["identical"] for r in recs) else "DIFFERENCES")
sys.exit(0 if all(r["identical"] for r in recs) else 1)
