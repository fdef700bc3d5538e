#!/usr/bin/env python3
"""Developer probe for the GPU box: primitives, full-path parity on one fixture, stage timings
on synthetic read sets.  Writes gpurun_out/gpu_check.json.  (The judged artefacts are the
pytest -m gpu suite and bench.py; this script is for quick iteration.)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd import _host
from centroflye_amd.engine import Engine


def log(*a):
    print(*a, flush=True)


def parity(e, name):
    import fixtures
    from oracle import ncrf, recruit, placer
    os.makedirs("/tmp/fx", exist_ok=True)
    rp = fixtures.make_report(name, "/tmp/fx")
    p2, p3 = fixtures.stage2_params(name), fixtures.stage3_params(name)
    records, alns, lens = ncrf.parse_report(rp)
    res = recruit.stage2(records, k=p2["k"], coverage=p2["coverage"], min_coverage=p2["min_coverage"], max_d=p2["max_distance"])
    pk = _host.parse_report(rp)
    out = {}
    e.load(pk, 1)
    e.count_kmers(p2["k"])
    keys, pres, multi = e.table()
    ok = multi <= 3
    out["A1"] = bool(np.array_equal(keys[ok], res["keys"]) and np.array_equal(pres[ok].astype(np.int64), res["pres"]))
    lo, hi = recruit.rare_bounds(0.9, 3.0, p2["coverage"], 0.34)
    e.select_rare(3, lo, hi)
    out["A2"] = bool(np.array_equal(e.kmers(), res["rare"]))
    e.build_clouds()
    cp, ent = e.clouds()
    out["A3"] = bool(np.array_equal(cp, res["cloud_ptr"]) and np.array_equal(ent, res["entries"]))
    ne = e.dist_edges(0, 2 ** 62, 1, p2["max_distance"], p2["min_coverage"], 0.8, 0, 1, edge_cap=res["counters"]["n_edges"] + 10)
    ed = e.edges(ne).astype(np.int64)
    ed = ed[np.lexsort((ed[:, 2], ed[:, 1], ed[:, 0]))]
    st = e.stats()
    out["A5A6"] = bool(np.array_equal(ed, res["edges"]) and st["n_emissions"] == res["counters"]["E"]
                       and np.array_equal(np.flatnonzero(e.unique_mask()), res["unique"]))
    out["dist_kernel_ms"] = e.times()["dist_kernel_ms"]
    out["E"] = st["n_emissions"]
    gk = res["rare"][res["unique"]]
    r3 = placer.stage3(records, alns, lens, gk, min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    e.set_kmers(gk, p3["k_cloud"]); e.build_clouds(); e.filter_clouds(p3["min_kmer_mult"])
    cp, ent = e.clouds()
    out["A4"] = bool(np.array_equal(cp, r3["f_cloud_ptr"]) and np.array_equal(ent, r3["f_entries"]))
    cls = pk.classify(p3["prefix_threshold"])
    ids = pk.ids
    rank = np.argsort(np.argsort(np.array(ids))).astype(np.int32)
    rd, pos, s0, s1 = e.place_reads(cls, rank, p3["min_cloud_kmer_freq"], p3["min_unit"], p3["min_inters"], 3)
    lines = []
    for a, b, c, d in zip(rd, pos, s0, s1):
        lines.append(f"{ids[a]} 0" if (c < 0 and b == 0) else (f"{ids[a]} None" if b < 0 else f"{ids[a]} {b} {c} {d}"))
    out["A9"] = lines == r3["lines"]
    out["place_ms"] = e.times()["place_ms"]
    return out


def timing(e, n_reads, seed, place=True):
    n_units = max(30, int(n_reads * 20000 * 0.93 / 32 / 2055))
    t = time.time()
    pk = _host.synth(seed=seed, n_units=n_units, n_reads=n_reads)
    t_synth = time.time() - t
    out = dict(n_reads=pk.n_reads, n_bases=pk.n_bases, n_units_array=n_units, synth_s=round(t_synth, 2))
    t0 = time.time(); e.load(pk, 1); out["load_s"] = round(time.time() - t0, 3)
    t0 = time.time(); e.count_kmers(19); out["count_s"] = round(time.time() - t0, 3)
    t0 = time.time(); nr = e.select_rare(3, 10, 32); out["select_s"] = round(time.time() - t0, 3)
    t0 = time.time(); e.build_clouds(); out["clouds_s"] = round(time.time() - t0, 3)
    t0 = time.time(); ne = e.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=0); out["dist_s"] = round(time.time() - t0, 3)
    out["stats"] = e.stats(); out["times"] = e.times()
    tot = out["count_s"] + out["select_s"] + out["clouds_s"] + out["dist_s"]
    out["bases_per_s_recruit_dist"] = pk.n_bases / tot
    if place:
        mask = e.unique_mask(); gk = e.kmers()[mask]
        e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
        cls = pk.classify(50000)
        rank = np.argsort(np.argsort(np.array(pk.ids))).astype(np.int32)
        t0 = time.time(); rd, pos, s0, s1 = e.place_reads(cls, rank, 2, 2, 10, 3); out["place_s"] = round(time.time() - t0, 3)
        out["placed"] = int((pos >= 0).sum()); out["none"] = int((pos < 0).sum())
        out["classes"] = np.bincount(cls, minlength=3).tolist()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1000")
    ap.add_argument("--fixture", default="lowcov")
    ap.add_argument("--no-place", action="store_true")
    ap.add_argument("--param", action="append", default=[])
    a = ap.parse_args()
    res = {}
    e = Engine(0)
    res["device"] = e.device_info(); log(res["device"])
    for p in a.param:
        k, v = p.split("="); e.set_param(k, int(v))
    rng = np.random.default_rng(1)
    for n in (0, 1, 2049, 100000, 5000000):
        v = rng.integers(0, 1000, n)
        assert (e.selftest_scan(v) == np.concatenate([[0], np.cumsum(v)])).all(), ("scan", n)
        k = rng.integers(0, 2 ** 38, n, dtype=np.uint64)
        assert (e.selftest_sort(k, 38) == np.sort(k)).all(), ("sort", n)
    log("prims ok"); res["prims"] = True
    if a.fixture:
        res["parity"] = parity(e, a.fixture); log("parity", res["parity"])
    res["timing"] = []
    for i, n in enumerate(int(x) for x in a.sizes.split(",") if x):
        r = timing(e, n, 100 + i, place=not a.no_place); log(json.dumps(r)); res["timing"].append(r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gpu_check.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
