#!/usr/bin/env python3
"""BASELINE configs[2] at its own size, once: the distance stage (A5 + A6) of the bench workload — 50 000 reads,
1.5e11 pair emissions — on the CPU (oracle/c/cf_oracle_mt.c, all host cores), as 64 partitions of the first k-mers
(a % 64 == p), summed; then the GPU's full launch on the same reads, compared.  Writes gpurun_out/r03_full_parity.json;
the committed copy under profiles/ is what tests/test_gpu_fullsize.py and bench.py assert the GPU result against on
every run.  Runs ON THE GPU BOX (≈ 7 min of CPU).  Reference: distance_based_kmer_recruitment.py:85-149.
usage: python3 tools/full_parity.py [--reads 50000] [--seed 2] [--parts 64] [--out gpurun_out/r03_full_parity.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from centroflye_amd import _host  # noqa: E402
from oracle import cport  # noqa: E402

P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=50000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--parts", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r03_full_parity.json"))
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    pk = _host.synth(n_reads=a.reads, seed=a.seed, n_units=max(24, int(round(0.3 * a.reads))), var_len=8)
    up, us, ue, _ = pk.units(1)
    t0 = time.time()
    st = cport.Stage2State(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0)
    c = st.counters
    res = dict(workload=dict(reads=a.reads, seed=a.seed, n_units=max(24, int(round(0.3 * a.reads))), var_len=8, params=P),
               source="oracle/c/cf_oracle_mt.c (cfo_mt_prepare + cfo_mt_dist_part), %d partitions of the first k-mers, %d host threads" % (a.parts, os.cpu_count()),
               n_bases=int(pk.n_bases), n_rare=c["n_rare"], n_cloud_entries=c["n_cloud_entries"], rare_checksum=c["rare_checksum"],
               cloud_checksum=c["cloud_checksum"], table_checksum=c["table_checksum"], n_distinct=c["n_distinct"], n_kept=c["n_kept"],
               n_windows=c["n_windows"], n_read_kmers=c["n_read_kmers"],
               cpu_secs=dict(count_select=st.secs_count_select, clouds=st.secs_clouds, dist_parts=[]), parts=[])
    uq = np.zeros(c["n_rare"], np.uint8)
    em = ne = chk = 0
    for p in range(a.parts):
        w = st.dist_part(p, a.parts, 0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], threads=0, unique=uq)
        em += w["n_emissions"]; ne += w["n_edges"]; chk = (chk + w["edge_checksum"]) % 2 ** 64
        res["parts"].append(dict(part=p, n_emissions=w["n_emissions"], n_edges=w["n_edges"], edge_checksum=w["edge_checksum"], n_unique=w["n_unique"]))
        res["cpu_secs"]["dist_parts"].append(round(w["secs"], 3))
        res.update(n_emissions=em, n_edges=ne, edge_checksum=chk, n_unique=int(uq.sum()), parts_done=p + 1)
        print(f"part {p}: {w['secs']:.1f} s, {w['n_emissions']} emissions, {w['n_edges']} edges (total wall {time.time() - t0:.0f} s)", flush=True)
        with open(a.out + ".partial", "w") as f:
            json.dump(res, f)
    rare = st.arrays()["rare"]
    res["unique_kmers_checksum"] = cport.rare_checksum(rare[uq.astype(bool)])
    res["cpu_secs"]["dist_total"] = round(sum(res["cpu_secs"]["dist_parts"]), 2)
    res["cpu_total_secs"] = round(st.secs_count_select + st.secs_clouds + res["cpu_secs"]["dist_total"], 2)
    res["cpu_bases_per_s"] = pk.n_bases / res["cpu_total_secs"]
    res["cpu_emissions_per_s"] = em / res["cpu_secs"]["dist_total"]
    res["host_cpus"] = os.cpu_count()
    st.close()
    if not a.no_gpu:
        from centroflye_amd.engine import Engine
        with Engine(0) as e:
            e.load(pk, 1)
            e.count_kmers(P["k"])
            n_rare = e.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
            n_ce = e.build_clouds()
            e.reset_unique()
            n = e.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=ne + 16)
            s = e.stats()
            g = dict(n_rare=n_rare, n_cloud_entries=n_ce, n_edges=n, n_emissions=s["n_emissions"], n_unique=s["n_unique"],
                     edge_checksum=e.edges_checksum(), unique_kmers_checksum=cport.rare_checksum(e.kmers()[e.unique_mask()]),
                     rare_equal=bool(np.array_equal(e.kmers(), rare)), unique_mask_equal=bool(np.array_equal(e.unique_mask(), uq.astype(bool))),
                     device=e.device_info()["name"].strip(), dist_kernel_ms=e.times()["dist_kernel_ms"])
        res["gpu"] = g
        res["gpu_equals_cpu"] = bool(all(g[k] == res[k] for k in ("n_rare", "n_cloud_entries", "n_edges", "n_emissions", "n_unique", "edge_checksum",
                                                                 "unique_kmers_checksum")) and g["rare_equal"] and g["unique_mask_equal"])
        print("GPU == CPU:", res["gpu_equals_cpu"], g, flush=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    try:
        os.remove(a.out + ".partial")
    except OSError:
        pass
    print(json.dumps({k: v for k, v in res.items() if k not in ("parts",)}))


if __name__ == "__main__":
    main()
