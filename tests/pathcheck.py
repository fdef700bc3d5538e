"""Shared parity checks: drive the device pipeline through the C ABI (centroflye_amd.engine.Engine)
and compare with the oracle.  Used by the CPU suite on the host-emulated build of the kernels
(small cases) and by the `-m gpu` suite on a real MI355X (all cases)."""
import numpy as np

from centroflye_amd import _host
from conftest import lines_from_placement
from oracle import placer, recruit


def sorted_edges(e):
    e = np.asarray(e).astype(np.int64).reshape(-1, 4)
    return e[np.lexsort((e[:, 2], e[:, 1], e[:, 0]))]


def check_stage2(engine, report_path, oracle_tuple, n_parts=1, check_table=True):
    """A1..A6 of one fixture; oracle_tuple from the oracle_stage2 fixture."""
    records, alns, lens, res, p2 = oracle_tuple
    pk = _host.parse_report(report_path)
    engine.load(pk, 1)
    engine.count_kmers(p2["k"])
    st = engine.stats()
    cn = res["counters"]
    assert (st["n_bases"], st["n_windows"], st["n_read_kmers"]) == (cn["n_b"], cn["n_w"], cn["n_rk"])
    if check_table:
        keys, pres, multi = engine.table()
        assert keys.size == cn["n_distinct"]
        ok = multi <= p2["max_nonuniq"]
        assert np.array_equal(keys[ok], res["keys"]), "A1 keys"
        assert np.array_equal(pres[ok].astype(np.int64), res["pres"]), "A1 presence counts"
    n = engine.select_rare(p2["max_nonuniq"], cn["lo"], cn["hi"])
    assert n == res["rare"].size
    assert np.array_equal(engine.kmers(), res["rare"]), "A2 rare set (sorted)"
    assert engine.stats()["n_distinct"] == cn["n_distinct"]
    nce = engine.build_clouds()
    cp, ent = engine.clouds()
    assert nce == res["entries"].size
    assert np.array_equal(cp, res["cloud_ptr"]) and np.array_equal(ent, res["entries"]), "A3 clouds"
    engine.reset_unique()
    parts = []
    E = 0
    for part in range(n_parts):
        ne = engine.dist_edges(p2["min_nreads"], p2["max_nreads"], p2["min_distance"], p2["max_distance"],
                               p2["min_coverage"], 0.8, part, n_parts, edge_cap=res["edges"].shape[0] + 8)
        parts.append(engine.edges(ne))
        from oracle import cport
        assert engine.edges_checksum() == cport.edge_checksum(parts[-1]), "cf_edges_checksum (device) vs the rows"
        assert engine.edges_checksum(ne // 2) == cport.edge_checksum(parts[-1][:ne // 2])
        engine.sort_edges()                  # the device's own (d, a, b) order = numpy's lexsort of the same rows
        assert np.array_equal(engine.edges(ne), sorted_edges(parts[-1])), "cf_sort_edges"
        E += engine.stats()["n_emissions"]
    ed = sorted_edges(np.concatenate(parts))
    assert E == cn["E"], "pair emissions"
    assert np.array_equal(ed, res["edges"]), "A5+A6 selected edges"
    assert np.array_equal(np.flatnonzero(engine.unique_mask()), res["unique"]), "A6 unique k-mers"
    assert engine.stats()["n_unique"] == res["unique"].size
    return pk


def check_stage3(engine, pk, records, alns, lens, genomic_kmers, p3, expect_lines=None):
    """A3 (placer k-mer set), A4, A8, A9."""
    r3 = placer.stage3(records, alns, lens, genomic_kmers, n_motif=p3["n_motif"], k_cloud=p3["k_cloud"],
                       min_cloud_kmer_freq=p3["min_cloud_kmer_freq"], min_kmer_mult=p3["min_kmer_mult"],
                       min_unit=p3["min_unit"], min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    engine.load(pk, p3["n_motif"])
    engine.set_kmers(genomic_kmers, p3["k_cloud"])
    engine.build_clouds()
    cp, ent = engine.clouds()
    assert np.array_equal(cp, r3["cloud_ptr"]) and np.array_equal(ent, r3["entries"]), "A3 clouds (placer set)"
    engine.filter_clouds(p3["min_kmer_mult"])
    cp, ent = engine.clouds()
    assert np.array_equal(cp, r3["f_cloud_ptr"]) and np.array_equal(ent, r3["f_entries"]), "A4 filtered clouds"
    cls = pk.classify(p3["prefix_threshold"])
    assert np.array_equal(cls.astype(np.int64), r3["classes"])
    rank = np.argsort(np.argsort(np.array(pk.ids))).astype(np.int32)
    rd, pos, s0, s1 = engine.place_reads(cls, rank, p3["min_cloud_kmer_freq"], p3["min_unit"], p3["min_inters"], 3)
    lines = lines_from_placement(pk.ids, rd, pos, s0, s1)
    assert lines == r3["lines"], "A9 read_positions.csv lines"
    if expect_lines is not None:
        placed = [ln for ln in lines if not ln.endswith(" None")]
        none = sorted(ln for ln in lines if ln.endswith(" None"))
        assert placed == expect_lines["placed"] and none == expect_lines["none"], "A9 vs reference golden"
    return lines


def check_synthetic_clouds(engine, n_reads=3, n_units=100, cloud=6, n_kmers=40, seed=0, min_d=1, max_d=150, min_cov=2, kmer_base=0):
    """Distance stage on hand-made clouds (cf_set_clouds): k-mer 0 sits in EVERY unit, so its posting list
    (n_reads * n_units entries) exceeds the kernel's per-chunk posting capacity and the multi-chunk path runs.
    kmer_base > 0: the other k-mers get the ranks kmer_base + 1 .. (a set of kmer_base + n_kmers k-mers: ranks beyond 24 bits)."""
    rng = np.random.default_rng(seed)
    unit_ptr = np.arange(n_reads + 1, dtype=np.int64) * n_units
    U = n_reads * n_units
    ent, cp = [], [0]
    for _ in range(U):
        c = np.unique(np.concatenate([[0], rng.integers(1, n_kmers, cloud - 1)])).astype(np.int32)
        ent.append(c)
        cp.append(cp[-1] + c.size)
    entries, cloud_ptr = np.concatenate(ent), np.array(cp, np.int64)
    if kmer_base:
        entries = np.where(entries > 0, entries + kmer_base, 0).astype(np.int32)
        n_kmers += kmer_base
    a, b, d, cnt, E = recruit.dist_histogram(unit_ptr, cloud_ptr, entries, n_kmers, 0, n_reads, min_d, max_d)
    edges, uniq = recruit.filter_edges(a, b, d, cnt, min_cov)
    zeros = np.zeros(U, np.int64)
    engine.load_arrays(np.zeros(0, np.uint8), np.zeros(n_reads + 1, np.int64), unit_ptr, zeros, zeros)
    engine.set_kmers(np.arange(n_kmers, dtype=np.uint64), 19)
    engine.set_clouds(cloud_ptr, entries)
    engine.reset_unique()
    ne = engine.dist_edges(0, n_reads, min_d, max_d, min_cov, 0.8, 0, 1, edge_cap=edges.shape[0] + 8)
    assert engine.stats()["n_emissions"] == E
    assert np.array_equal(sorted_edges(engine.edges(ne)), edges)
    assert np.array_equal(np.flatnonzero(engine.unique_mask()), uniq)


def check_unit_kmers(engine, report_path, golden_entry, k):
    """occurrence table and top-n k-mers against the oracle and the reference golden."""
    import canon
    from oracle import ncrf, unit_kmers
    records, _, _ = ncrf.parse_report(report_path)
    recs = list(records.values())
    seqs = [r.r_al.replace("-", "").encode() for r in recs]
    okeys, ocnt = unit_kmers.kmer_occurrences(seqs, k)
    pk = _host.parse_report(report_path)
    engine.load(pk, 1)
    engine.count_occurrences(k)
    keys, lo, hi = engine.table()
    cnt = lo.astype(np.int64) | (hi.astype(np.int64) << 32)
    assert np.array_equal(keys, okeys) and np.array_equal(cnt, ocnt), "occurrence counts"
    g = golden_entry["k"][str(k)]
    assert keys.size == g["n_distinct"] and int(cnt.sum()) == g["total"]
    n = 3 * unit_kmers.n_circular_unit_kmers(recs[0].motif, k)
    tk, tc = engine.top_kmers(n)
    want = unit_kmers.most_frequent(okeys, ocnt, n)
    assert np.array_equal(tk, okeys[want]) and np.array_equal(tc.astype(np.int64), ocnt[want]), "top-n k-mers"
    strs = [recruit.decode_kmer(c, k) for c in tk]
    assert len(strs) == g["n_top"] and canon.set_digest(strs) == g["top_digest"], "top-n vs reference golden"
    assert [[s_, int(c_)] for s_, c_ in zip(strs[:20], tc[:20])] == g["top_head"]
    assert [[s_, int(c_)] for s_, c_ in zip(strs[-5:], tc[-5:])] == g["top_tail"]
    # degenerate requests
    assert engine.top_kmers(0)[0].size == 0
    allk, allc = engine.top_kmers(10 ** 9)
    assert allk.size == okeys.size and int(allc.astype(np.int64).sum()) == g["total"]


def check_argmax_selftest(engine):
    """The placement's candidate reduction against the host (reference read_placer.py:63-78: larger (s0, s1), then the
    larger offset, then the smaller id).  Layouts that decide which lane takes whose candidate at which shuffle step —
    round 3 found the six-field struct form of this reduction miscompiled (a lane that took its partner's candidate at
    distance 32 and none afterwards kept its OLD read id next to the new scores)."""
    rng = np.random.default_rng(9)

    def best(c):
        idx = [i for i in range(c.shape[0]) if c[i, 4]]
        if not idx:
            return None
        return max(idx, key=lambda i: (int(c[i, 0]), int(c[i, 1]), int(c[i, 2]), -int(c[i, 3])))

    cases = []
    # the failing pattern: a worse candidate in lane L, the best one in lane L + 32 of the same wave, nothing else
    for wave in range(4):
        for lane in (0, 5, 31):
            c = np.zeros((256, 5), np.uint32)
            c[64 * wave + lane] = (6, 476, 5210, 35727, 1)
            c[64 * wave + lane + 32] = (8, 764, 5360, 31523, 1)
            cases.append(c)
    for d in (1, 2, 4, 8, 16, 32):      # ... and at every other shuffle distance, in both orders
        for swap in (0, 1):
            c = np.zeros((256, 5), np.uint32)
            a, b = (70, 70 + d) if not swap else (70 + d, 70)
            c[a] = (3, 90, 17, 5, 1); c[b] = (3, 90, 17, 4, 1)      # a tie down to the id rank
            cases.append(c)
    for n in (0, 1, 63, 64, 65, 255, 256, 257, 1000, 5000):      # random sets with heavy ties, some invalid rows
        c = np.zeros((n, 5), np.uint32)
        c[:, 0] = rng.integers(0, 3, n); c[:, 1] = rng.integers(0, 4, n); c[:, 2] = rng.integers(0, 3, n)
        c[:, 3] = rng.permutation(n); c[:, 4] = rng.integers(0, 4, n) > 0
        cases.append(c)
    cases.append(np.zeros((300, 5), np.uint32))      # no valid candidate
    for c in cases:
        got = engine.selftest_argmax(c)
        w = best(c)
        if w is None:
            assert got[5] == 0, got
        else:
            assert got.tolist() == [int(c[w, 0]), int(c[w, 1]), int(c[w, 2]), int(c[w, 3]), w, 1], (got.tolist(), w, c[w].tolist())
