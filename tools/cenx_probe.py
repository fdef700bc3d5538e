#!/usr/bin/env python3
"""Developer tool (GPU box): stage 2 (+ optionally placement) on a cenX-SHAPED read set — BASELINE configs[4]'s regime: a short array
(1 500 units), coverage 32, ultra-long reads (mean 100 kb: ~47 units per read, distances up to max_d = 150 all occur) — per knob setting.
usage: tools/cenx_probe.py [--var-len 8] [--reads 1000] [--units 1500] [--mean-len 100000] [--lib path] [knobs: name=value,... ]...
Prints per setting: stage ms, kernel ms of two launches, first k-mers, passes, spilled, pair emissions, emissions/s, edges.
Reference: distance_based_kmer_recruitment.py:85-149 (max_d 150 exists because reads span 150 units), run_all_cenX.sh:17-22."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--var-len", type=int, default=8); ap.add_argument("--reads", type=int, default=1000); ap.add_argument("--units", type=int, default=1500)
ap.add_argument("--mean-len", type=float, default=100000.0); ap.add_argument("--max-len", type=int, default=1000000); ap.add_argument("--seed", type=int, default=5)
ap.add_argument("--lib", default=None); ap.add_argument("--edges", action="store_true", help="store every selected edge (a sizing launch first)")
ap.add_argument("--place", action="store_true")
ap.add_argument("--once", type=int, default=0, help="ONE launch of the distance stage with this edge cap (the PMC passes of tools/profile_round.sh: every kernel runs once)")
ap.add_argument("knobs", nargs="*")
a = ap.parse_args()
pk = _host.synth(n_reads=a.reads, seed=a.seed, n_units=a.units, var_len=a.var_len, mean_len=a.mean_len, max_len=a.max_len)
print(f"reads {pk.n_reads} bases {pk.n_bases}", flush=True)
for spec in (a.knobs or [""]):
    e = Engine(0, _lib.load(os.path.join(ROOT, a.lib)) if a.lib else None)
    for kv in filter(None, spec.split(",")):
        e.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    e.load(pk, 1); e.count_kmers(19); n_rare = e.select_rare(3, 10, 32); n_ce = e.build_clouds()
    tm = e.times()
    ms, cap = [], a.once
    for i in range(1 if a.once else 3):
        n_edges = e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, cap)
        ms.append(round(e.times()["dist_kernel_ms"], 2))
        if a.edges and i == 0:
            cap = n_edges + 1024
    st = e.stats()
    rec = dict(knobs=spec, n_rare=n_rare, n_cloud_entries=n_ce, count_ms=round(tm["count_ms"], 2), select_ms=round(tm["select_ms"], 2), clouds_ms=round(tm["clouds_ms"], 2),
               setup_ms=round(e.times()["postings_ms"], 2), kernel_ms=ms, n_emissions=st["n_emissions"], n_edges=n_edges, n_unique=st["n_unique"], passes=st["n_dist_passes"],
               spilled=st["n_spilled"], emissions_per_s=st["n_emissions"] / (min(ms[1:] or ms) * 1e-3), per_first=st["n_emissions"] / max(n_rare, 1))
    if a.place:
        import numpy as np
        gk = e.kmers()[e.unique_mask()]
        e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
        cls = pk.classify(50000)
        idr = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
        t0 = time.perf_counter()
        rd, pos, s0, s1 = e.place_reads(cls, idr, 2, 2, 10, 3)
        rec.update(place_s=round(time.perf_counter() - t0, 3), place_device_ms=round(e.times()["place_ms"], 1), placed=int((pos >= 0).sum()))
    print(json.dumps(rec), flush=True)
    e.close()
