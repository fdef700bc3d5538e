"""A1 + A2 + A3 and ONE first-k-mer partition of A5/A6 of a large synthetic read set on the GPU against the OpenMP oracle
(oracle/c/cf_oracle_mt.c) — what tests/test_gpu_fullsize.py does at 50 000 reads, at sizes whose k-mer sets take the other
table layouts of the distance kernel and three bucket passes of the counting sort.  Used by the -m gpu test at 200 000
reads (BASELINE configs[3]'s single-GPU-feasible share) and by tools/rank_emulation.py --check at 400 000.
Reference: distance_based_kmer_recruitment.py:39-149 (dist_cnt[d][a] is a's own dict, :108-113: a partition of the first
k-mers is an independent piece of the same result)."""
import time

import numpy as np

from oracle import cport

P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)


def check(engine, pk, part, n_parts, loaded=False):
    """Returns a record (counters, checksums, seconds, `identical`).  The engine is left with the read set loaded, the rare
    set selected and the clouds built."""
    up, us, ue, _ = pk.units(1)
    if not loaded:
        engine.load(pk, 1)
    engine.count_kmers(P["k"])
    keys, pres, multi = engine.table(sort=False)
    tchk, n_table = cport.table_checksum(keys, pres, multi), int(keys.size)
    del keys, pres, multi
    n_rare = engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
    st = engine.stats()
    n_ce = engine.build_clouds()
    rec = dict(reads=int(pk.n_reads), n_bases=int(st["n_bases"]), n_windows=int(st["n_windows"]), n_read_kmers=int(st["n_read_kmers"]), n_distinct=int(st["n_distinct"]),
               n_kept=int(st["n_kept"]), n_rare=int(n_rare), n_cloud_entries=int(n_ce), table_checksum=int(tchk), partition=f"a % {n_parts} == {part}")
    t0 = time.time()
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0) as st2:
        rec["oracle_A1_A3_s"] = round(time.time() - t0, 1)
        c, a = st2.counters, st2.arrays()
        ok = {}
        ok["counters"] = (st["n_bases"], st["n_windows"], st["n_read_kmers"], st["n_distinct"], n_table, st["n_kept"], n_rare, n_ce) == \
                         (c["n_bases"], c["n_windows"], c["n_read_kmers"], c["n_distinct"], c["n_distinct"], c["n_kept"], c["n_rare"], c["n_cloud_entries"])
        ok["table_checksum"] = tchk == c["table_checksum"]
        ok["rare_set"] = bool(np.array_equal(engine.kmers(), a["rare"]))
        cp, ent = engine.clouds()
        ok["clouds"] = bool(np.array_equal(cp, a["cloud_ptr"]) and np.array_equal(ent, a["entries"]))
        del cp, ent, a
        uq = np.zeros(n_rare, np.uint8)
        t0 = time.time()
        w = st2.dist_part(part, n_parts, 0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], threads=0, unique=uq)
        rec["oracle_partition_s"] = round(time.time() - t0, 1)
    engine.reset_unique()
    ne = engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], part, n_parts, edge_cap=w["n_edges"] + 16)
    s2 = engine.stats()
    ok["partition_counters"] = (ne, s2["n_emissions"], s2["n_unique"]) == (w["n_edges"], w["n_emissions"], w["n_unique"])
    ok["edge_checksum"] = engine.edges_checksum() == w["edge_checksum"]
    ok["unique_bits"] = bool(np.array_equal(engine.unique_mask(), uq.astype(bool)))
    rec.update(n_emissions_partition=int(s2["n_emissions"]), n_edges_partition=int(ne), n_unique_partition=int(s2["n_unique"]), edge_checksum=int(w["edge_checksum"]),
               dist_kernel_ms=round(float(engine.times()["dist_kernel_ms"]), 1), n_dist_passes=int(s2["n_dist_passes"]), checks=ok, identical=all(ok.values()))
    return rec
