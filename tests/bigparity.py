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


# ---------------------------------------------------------------------------------------------------------------------
# Committed oracle records (round 5): the oracle side of a full-size check is run ONCE on the GPU box's host cores by
# tools/parity_record.py and committed under profiles/ (workload, counters, checksums of the A1 table, the rare set, the
# cloud CSR, and of one first-k-mer partition: pair emissions, edges, edge checksum, unique k-mers); the -m gpu tests then
# compare the device-side checksums (cf_checksum, cf_edges_checksum) of the same seeded reads with the record — every
# element of BASELINE configs[3]'s 500 000 reads is covered without 20 minutes of CPU per test run.

def synth_workload(wl):
    """The read set a record was taken on (generator parameters are part of the record)."""
    from centroflye_amd import _host
    kw = dict(wl.get("synth", {}))
    return _host.synth(n_reads=wl["reads"], seed=wl["seed"], n_units=wl["n_units"], var_len=wl["var_len"], **kw)


def oracle_record(pk, part, n_parts, sub=1, progress=None):
    """A1-A3 whole + the first k-mers a % n_parts == part (run as `sub` sub-partitions a % (n_parts * sub) == part + j * n_parts,
    which bounds the oracle's memory) on every host core."""
    up, us, ue, _ = pk.units(1)
    rec = dict(params=dict(P), reads=int(pk.n_reads))
    t0 = time.time()
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0) as st:
        c = st.counters
        rec.update({k: int(c[k]) for k in ("n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_units", "n_cloud_entries",
                                            "table_checksum", "rare_checksum", "cloud_checksum")})
        rec["oracle_A1_A3_s"] = round(time.time() - t0, 1)
        if progress:
            progress(rec)
        uq = np.zeros(c["n_rare"], np.uint8)
        tot = dict(n_emissions=0, n_edges=0, edge_checksum=0, n_first_kmers=0)
        t0 = time.time()
        for j in range(sub):
            w = st.dist_part(part + j * n_parts, n_parts * sub, 0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], threads=0, unique=uq)
            tot["n_emissions"] += int(w["n_emissions"]); tot["n_edges"] += int(w["n_edges"]); tot["n_first_kmers"] += int(w["n_first_kmers"])
            tot["edge_checksum"] = (tot["edge_checksum"] + int(w["edge_checksum"])) & (2 ** 64 - 1)
            if progress:
                progress(dict(rec, partial=dict(tot, done=j + 1, of=sub, secs=round(time.time() - t0, 1))))
        rare = st.arrays()["rare"]
    rec["partition"] = dict(tot, part=int(part), n_parts=int(n_parts), sub=int(sub), n_unique=int(uq.sum()),
                            unique_kmers_checksum=int(cport.rare_checksum(rare[uq.astype(bool)])), oracle_s=round(time.time() - t0, 1))
    return rec


def check_record(engine, pk, rec, loaded=False, through_exchange=False, rendezvous=None):
    """The GPU against a committed oracle record.  through_exchange: the partition is run the way one rank of n_parts runs it
    (tools/rank_emulation.py): A1 on the rank's read shard, the table exchange / rare-list gather / cloud gather through a
    one-rank communicator with comm_self_p2p (bucketing, ncclSend / ncclRecv rounds to itself, merge, gathered view), then A5/A6
    for a % n_parts == part over the gathered view with the union rare set installed."""
    p, part = rec["params"], rec["partition"]
    if not loaded:
        engine.load(pk, 1)
    ok, got = {}, {}
    engine.count_kmers(p["k"])
    st = engine.stats()
    tchk, n_table = engine.checksum("table")
    n_rare = engine.select_rare(p["max_nonuniq"], p["lo"], p["hi"])
    st2 = engine.stats()
    rchk, n_r = engine.checksum("kmers")
    n_ce = engine.build_clouds()
    cchk, n_c = engine.checksum("clouds")
    got.update(n_bases=st["n_bases"], n_windows=st["n_windows"], n_read_kmers=st["n_read_kmers"], n_distinct=st2["n_distinct"], n_kept=st2["n_kept"],
               n_rare=n_rare, n_cloud_entries=n_ce, table_checksum=tchk, rare_checksum=rchk, cloud_checksum=cchk)
    ok["counters"] = all(got[k] == rec[k] for k in ("n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_cloud_entries")) \
        and (n_table, n_r, n_c) == (rec["n_distinct"], rec["n_rare"], rec["n_cloud_entries"])
    ok["table_checksum"] = tchk == rec["table_checksum"]
    ok["rare_checksum"] = rchk == rec["rare_checksum"]
    ok["cloud_checksum"] = cchk == rec["cloud_checksum"]
    secs = {}
    if through_exchange:
        import os
        n, r = part["n_parts"], part["part"]
        rare = engine.kmers().copy()                       # the union set: what the all-gather of every owner's list installs
        lo, hi = r * pk.n_reads // n, (r + 1) * pk.n_reads // n
        engine.set_param("comm_self_p2p", 1)
        engine.comm_init(0, 1, rendezvous or os.path.join(os.environ.get("TMPDIR", "/tmp"), f"cf_bigparity_{os.getpid()}.id"))
        try:
            t0 = time.perf_counter(); engine.count_kmers(p["k"], lo, hi); secs["count_shard"] = time.perf_counter() - t0
            t0 = time.perf_counter(); got["exchange_bytes"] = int(engine.exchange_table()); secs["table_exchange_self"] = time.perf_counter() - t0
            t0 = time.perf_counter(); engine.select_rare(p["max_nonuniq"], p["lo"], p["hi"]); engine.allgather_kmers(); secs["select_gather_self"] = time.perf_counter() - t0
            engine.set_kmers(rare, p["k"])
            t0 = time.perf_counter(); engine.build_clouds(); secs["clouds_all_reads"] = time.perf_counter() - t0
            ok["cloud_checksum_again"] = engine.checksum("clouds")[0] == rec["cloud_checksum"]
            t0 = time.perf_counter(); got["gathered_cloud_entries"] = int(engine.allgather_clouds()); secs["cloud_gather_self"] = time.perf_counter() - t0
            ok["gathered_view"] = got["gathered_cloud_entries"] == rec["n_cloud_entries"]
            engine.reset_unique()
            t0 = time.perf_counter()
            ne = engine.dist_edges(0, 2 ** 62, p["min_d"], p["max_d"], p["min_cov"], p["rel_threshold"], r, n, edge_cap=part["n_edges"] + 16)
            secs["dist_part"] = time.perf_counter() - t0
            s3 = engine.stats()
            echk = engine.edges_checksum()
            uchk, n_u = engine.checksum("unique")
        finally:
            engine.comm_free()
            engine.set_param("comm_self_p2p", 0)
    else:
        engine.reset_unique()
        t0 = time.perf_counter()
        ne = engine.dist_edges(0, 2 ** 62, p["min_d"], p["max_d"], p["min_cov"], p["rel_threshold"], part["part"], part["n_parts"], edge_cap=part["n_edges"] + 16)
        secs["dist_part"] = time.perf_counter() - t0
        s3 = engine.stats()
        echk = engine.edges_checksum()
        uchk, n_u = engine.checksum("unique")
    got.update(n_emissions_partition=s3["n_emissions"], n_edges_partition=ne, n_unique_partition=n_u, edge_checksum=echk, unique_kmers_checksum=uchk,
               n_dist_passes=s3["n_dist_passes"], dist_kernel_ms=round(float(engine.times()["dist_kernel_ms"]), 1))
    ok["partition_counters"] = (ne, s3["n_emissions"], n_u, s3["n_unique"]) == (part["n_edges"], part["n_emissions"], part["n_unique"], part["n_unique"])
    ok["edge_checksum"] = echk == part["edge_checksum"]
    ok["unique_kmers_checksum"] = uchk == part["unique_kmers_checksum"]
    return dict(got={k: (int(v) if not isinstance(v, float) else v) for k, v in got.items()}, secs={k: round(v, 3) for k, v in secs.items()},
                checks=ok, identical=all(ok.values()))
