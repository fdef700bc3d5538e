"""Parity of the HIP path with the oracle and with the reference goldens, through the C ABI, on
a real MI355X.  Bit-exact: everything on this path is integer / byte / index work (the one
floating-point test, cnt / total >= 0.8, is done in IEEE doubles on both sides)."""
import os

import numpy as np
import pytest

import fixtures
import pathcheck
from centroflye_amd import _host
from centroflye_amd.engine import DeviceError, Engine
from oracle import cport, ncrf, recruit

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(fixtures.FIXTURES)


@pytest.fixture(scope="module")
def engine():
    e = Engine(0)   # raises if libcfhip.so or the GPU is missing: no fallback
    yield e
    e.close()


def test_device_is_gfx950(engine):
    info = engine.device_info()
    assert "gfx950" in info["name"], info
    assert info["n_cu"] == 256


def test_placement_argmax_reduction(engine):
    """cf_selftest_argmax: the candidate reduction of the greedy placement on adversarial layouts (see pathcheck)."""
    pathcheck.check_argmax_selftest(engine)


def test_scan_and_sort(engine):
    rng = np.random.default_rng(1)
    for n in (0, 1, 63, 2048, 2049, 100003, 3_000_001):
        v = rng.integers(0, 1000, n)
        assert np.array_equal(engine.selftest_scan(v), np.concatenate([[0], np.cumsum(v)]))
        k = rng.integers(0, 2 ** 62, n, dtype=np.uint64)
        assert np.array_equal(engine.selftest_sort(k, 62), np.sort(k))


@pytest.mark.parametrize("name", NAMES)
def test_stage2_fixture(engine, name, report, oracle_stage2):
    pathcheck.check_stage2(engine, report(name), oracle_stage2(name))


@pytest.mark.parametrize("name", NAMES)
def test_stage3_fixture_against_reference_golden(engine, name, report, golden):
    g = golden(name)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name))
    pathcheck.check_stage3(engine, pk, records, alns, lens, gk, g["stage3"], expect_lines=g["read_positions"])


def test_wide_table_layout(engine, report, oracle_stage2):
    """The 8-byte-slot layout used for k-mer sets of 2^24 ranks or more (default: 6-byte slots)."""
    engine.set_param("dist_wide", 1)
    try:
        pathcheck.check_stage2(engine, report("tiny"), oracle_stage2("tiny"), check_table=False)
        engine.set_param("dist_slots", 512)
        engine.set_param("dist_sketch", 0)      # every (b, d) pair in the exact table: forces the spill path
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov"), n_parts=2, check_table=False)
        assert engine.stats()["n_spilled"] > 0
        engine.set_param("dist_sketch", 1)
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov"), n_parts=2, check_table=False)
        pathcheck.check_synthetic_clouds(engine, n_reads=4, n_units=150, cloud=8, n_kmers=60)
    finally:
        engine.set_param("dist_wide", 0)
        engine.set_param("dist_slots", 0)
        engine.set_param("dist_sketch", 1)


def test_partitions_spill_and_slices(engine, report, oracle_stage2):
    # first k-mers split 3 ways (the multi-GPU partition), tiny LDS table (forces the spill path)
    engine.set_param("dist_slots", 1024)
    engine.set_param("dist_stage", 5)           # with chunks of 32 rows of the edge output: late rows beyond the staged list -> marked-slot sweep
    engine.set_param("dist_edge_chunk", 32)
    try:
        engine.set_param("dist_sketch", 0)      # every (b, d) pair in the exact table
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov"), n_parts=3, check_table=False)
        assert engine.stats()["n_spilled"] > 0
        engine.set_param("dist_sketch", 1)      # counting sketch first (the default), same tiny LDS budget
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov"), n_parts=3, check_table=False)
        engine.set_param("dist_post_atomics", 1)    # the partition's postings by histogram + fill passes of atomics (default: compaction + sort)
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov"), n_parts=3, check_table=False)
        engine.set_param("dist_post_atomics", 0)
        engine.set_param("dist_slots", 256)     # 2048 8-bit counters: they wrap -> falls back to "every b marked"
        pathcheck.check_stage2(engine, report("hor2055"), oracle_stage2("hor2055"), check_table=False)
    finally:
        engine.set_param("dist_slots", 0)
        engine.set_param("dist_post_atomics", 0)
        engine.set_param("dist_stage", 2048)
        engine.set_param("dist_edge_chunk", 0)
        engine.set_param("dist_sketch", 1)
    # --min-nreads / --max-nreads slice and --min-distance 0 (kmer_clouds[:-0] is empty) and a narrow d window
    for ov in (dict(min_nreads=3, max_nreads=11), dict(min_distance=0, max_distance=4), dict(min_distance=2, max_distance=3)):
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov", **ov), check_table=False)


@pytest.mark.parametrize("name,k", [("tiny", 30), ("hor2055", 30), ("lowcov", 19)])
def test_unit_kmer_occurrences_and_top_n(engine, report, name, k):
    import json
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unit_kmers.json")) as f:
        g = json.load(f)
    pathcheck.check_unit_kmers(engine, report(name), g, k)
    engine.set_param("count_mode", 0)      # the atomic table of round 1 instead of sort and reduce: the same table and top n
    try:
        pathcheck.check_unit_kmers(engine, report(name), g, k)
    finally:
        engine.set_param("count_mode", 1)


def test_long_posting_lists_take_the_multi_chunk_path(engine):
    pathcheck.check_synthetic_clouds(engine, n_reads=4, n_units=150, cloud=8, n_kmers=60)   # 600 postings of k-mer 0
    pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=40, max_d=7, min_d=3, seed=5)


def test_max_distance_beyond_255_and_reads_of_many_units(engine):
    """Limits the reference does not have: --max-distance > 255 takes the 16-bit distance layout
    ([b:32 | d:16 | sel:1 | cnt:15]); unit indices travel mod 256 / mod 65536, so a read may have any number of units
    (here 70 000 units in one read, narrow layout, and 700 units with max_d 600, 16-bit layout)."""
    pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=700, cloud=4, n_kmers=40, max_d=600, seed=3)
    pathcheck.check_synthetic_clouds(engine, n_reads=1, n_units=70000, cloud=3, n_kmers=2000, max_d=3, seed=4)


def test_kmer_sets_beyond_2_pow_24_keep_the_six_byte_slots(engine):
    """[d : DB | b : 32 - DB] with DB < 8: 2^24 + 40 and 2^26 + 40 k-mers (ranks in the keys pass 24 / 26 bits) with reads of
    100 / 30 units; the same clouds through the 8-byte layout give the same edges (pathcheck compares both with the oracle)."""
    pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=100, cloud=4, n_kmers=40, seed=7, kmer_base=1 << 24)      # DB = 7
    pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=30, cloud=4, n_kmers=40, seed=8, kmer_base=1 << 26)       # DB = 5
    engine.set_param("dist_wide", 1)
    try:
        pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=100, cloud=4, n_kmers=40, seed=7, kmer_base=1 << 24)
    finally:
        engine.set_param("dist_wide", 0)
    # long reads (200 units: distances need 8 bits) with 2^24 + 40 and 2^26 + 40 k-mers: the region layout (2 and 8 regions)
    pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=200, cloud=4, n_kmers=40, seed=9, kmer_base=1 << 24)
    pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=200, cloud=4, n_kmers=40, seed=10, kmer_base=1 << 26)
    for regions, stream_bytes in ((1, 0), (4, 0), (4, 1)):      # (the 4-byte stream of round 4; rank and unit index apart)
        engine.set_param("dist_regions", regions); engine.set_param("dist_region_bytes", stream_bytes)
        try:
            pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=100, cloud=4, n_kmers=50, seed=regions)
            pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=128, cloud=5, n_kmers=30, max_d=127, seed=11 + regions)
            pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=100, cloud=40, n_kmers=300, max_d=99, seed=21 + regions)
        finally:
            engine.set_param("dist_regions", 0); engine.set_param("dist_region_bytes", 0)
    for dbits in (5, 6, 7):
        engine.set_param("dist_dbits", dbits)
        try:
            pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=min(100, (1 << dbits) - 2), cloud=4, n_kmers=50, seed=dbits)
        finally:
            engine.set_param("dist_dbits", 0)


def test_against_c_oracle_on_bench_like_sample(engine):
    """Same generator and parameters as the benchmark workload, at a size the C oracle finishes in
    ~20 s: every counter and the order-independent checksums of rare set, clouds and edges."""
    pk = _host.synth(seed=21, n_units=36, n_reads=130, var_len=8)
    up, us, ue, _ = pk.units(1)
    c, _ = cport.stage2(pk.bases, pk.read_off, up, us, ue, 19, 3, 10, 32, 0, 2 ** 62, 1, 150, 4, 0.8)
    engine.load(pk, 1)
    engine.count_kmers(19)
    assert engine.select_rare(3, 10, 32) == c["n_rare"]
    assert cport.rare_checksum(engine.kmers()) == c["rare_checksum"]
    assert engine.build_clouds() == c["n_cloud_entries"]
    cp, ent = engine.clouds()
    assert cport.cloud_checksum(cp, ent) == c["cloud_checksum"]
    ne = engine.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=c["n_edges"])
    st = engine.stats()
    assert (ne, st["n_emissions"], st["n_unique"]) == (c["n_edges"], c["n_emissions"], c["n_unique"])
    assert (st["n_windows"], st["n_read_kmers"], st["n_distinct"], st["n_kept"]) == (c["n_windows"], c["n_read_kmers"], c["n_distinct"], c["n_kept"])
    assert cport.edge_checksum(engine.edges(ne)) == c["edge_checksum"]


def test_full_size_properties(engine):
    """Half of BASELINE config-1 (500 reads, ~10 Mb, ~2.5e9 pair emissions): size-independent properties — the result does
    not depend on the LDS table size, on the partition of first k-mers, or on the run; storing
    fewer edges than selected changes nothing else."""
    pk = _host.synth(seed=1, n_units=150, n_reads=500, var_len=8)
    engine.load(pk, 1)
    engine.count_kmers(19)
    n_rare = engine.select_rare(3, 10, 32)
    assert n_rare > 5000
    rare = engine.kmers()
    assert np.all(rare[1:] > rare[:-1])                      # sorted, unique
    engine.build_clouds()
    cp, ent = engine.clouds()
    assert np.all(np.diff(cp) >= 0) and ent.min() >= 0 and ent.max() < n_rare
    inner = np.ones(ent.size, bool); inner[cp[1:-1][cp[1:-1] < ent.size]] = False
    assert np.all((np.diff(ent) > 0) | ~inner[1:])           # every cloud sorted-unique
    ref = None
    for slots, parts, sketch in ((0, 1, 1), (6000, 1, 0), (4096, 2, 1), (2048, 1, 1)):
        engine.set_param("dist_slots", slots)
        engine.set_param("dist_sketch", sketch)
        engine.reset_unique()
        tot_e = tot_n = 0
        chk = 0
        for p in range(parts):
            ne = engine.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, p, parts, edge_cap=60_000_000)
            assert ne <= 60_000_000
            chk = (chk + cport.edge_checksum(engine.edges(ne))) % 2 ** 64
            tot_e += engine.stats()["n_emissions"]; tot_n += ne
        got = (tot_n, tot_e, chk, engine.unique_mask().tobytes())
        if ref is None:
            ref = got
        assert got == ref
    engine.set_param("dist_slots", 0)
    engine.set_param("dist_sketch", 1)
    # launch shape (chosen by the library from the pair emissions per first k-mer): both shapes forced
    try:
        for block, wgs in ((1024, 1), (512, 2)):
            engine.set_param("dist_block", block); engine.set_param("dist_wgs", wgs)
            engine.reset_unique()
            ne = engine.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=60_000_000)
            assert (ne, engine.stats()["n_emissions"], cport.edge_checksum(engine.edges(ne)), engine.unique_mask().tobytes()) == ref
    finally:
        engine.set_param("dist_block", 0); engine.set_param("dist_wgs", 0)
    engine.reset_unique()
    ne = engine.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=60_000_000)
    full = {tuple(r) for r in engine.edges(ne).tolist()}
    for cap in (1000, 20000, ne - 1):      # count-only beyond the cap; the rows kept are whole, distinct edges of the full set (the chunks of
        engine.reset_unique()              # the output that workgroups reserve leave holes that are closed after the kernel)
        ne2 = engine.dist_edges(0, 2 ** 62, 1, 150, 4, 0.8, 0, 1, edge_cap=cap)
        stored = engine.stats()["n_edges_stored"]
        rows = [tuple(r) for r in engine.edges(stored).tolist()]
        assert ne2 == ref[0] and 0 < stored <= cap and len(set(rows)) == stored and set(rows) <= full, cap
        assert engine.unique_mask().tobytes() == ref[3]


def test_errors_and_degenerate_inputs(engine):
    engine.load_arrays(np.frombuffer(b"ACGTNACGTACGTACGTACGTACGTACGTAC", np.uint8), [0, 31], [0, 1], [0], [31])   # N: its windows are skipped
    engine.count_kmers(4)
    assert engine.table()[0].size == 4
    with pytest.raises(DeviceError, match="symbols"):
        engine.count_occurrences(4)
    engine.load_arrays(np.zeros(0, np.uint8), [0], [0], [], [])
    engine.count_kmers(19)
    assert engine.select_rare(3, 1, 10) == 0 and engine.build_clouds() == 0
    assert engine.dist_edges(0, 10, 1, 150, 1, 0.8) == 0
    seq = b"ACGTACGTACGTACGTACGTACGTAAAC" + b"ACGTA" + b"TTGACCA"
    engine.load_arrays(np.frombuffer(seq, np.uint8), [0, 28, 33, 40], [0, 2, 2, 3], [0, 20, 33], [20, 28, 40])
    engine.count_kmers(19)
    codes, cnt = np.unique(recruit.encode_windows(seq[:28], 19), return_counts=True)
    keys, pres, multi = engine.table()
    assert np.array_equal(keys, codes) and (pres == 1).all() and np.array_equal(multi, (cnt > 1).astype(np.uint32))
    assert engine.select_rare(0, 1, 1) == int((cnt == 1).sum())
    with pytest.raises(DeviceError, match="max_d"):
        engine.select_rare(3, 1, 1); engine.build_clouds(); engine.dist_edges(0, 10, 1, 70000, 1, 0.8)


def test_a_few_very_long_sequences_take_the_table_path(engine):
    """The reduce of the sort-and-reduce A1 looks back along a read's run of records inside a bucket; four 5-Mb sequences would
    make those runs hundreds of records long, so the host hands such input to the atomic table (cf_count2.hip:
    cf_count_sorted returns 1).  Same table either way: presence / multi against numpy, a repeated stretch included."""
    rng = np.random.default_rng(11)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    seqs = [alpha[rng.integers(0, 4, 5_200_000)] for _ in range(4)]
    seqs[1][1000:3000] = seqs[0][5000:7000]          # shared between two reads
    seqs[2][100_000:102_000] = seqs[2][500:2500]     # twice in one read
    bases = np.concatenate(seqs)
    off = np.concatenate([[0], np.cumsum([s.size for s in seqs])])
    engine.load_arrays(bases, off, np.zeros(len(seqs) + 1, np.int64), [], [])
    engine.count_kmers(19)
    keys, pres, multi = engine.table()
    per = [np.unique(recruit.encode_windows(s.tobytes(), 19), return_counts=True) for s in seqs]
    allk = np.concatenate([u for u, _ in per]); allm = np.concatenate([(c > 1) for _, c in per])
    o = np.argsort(allk, kind="stable")
    wk, start, wp = np.unique(allk[o], return_index=True, return_counts=True)
    wm = np.add.reduceat(allm[o].astype(np.int64), start)
    assert np.array_equal(keys, wk) and np.array_equal(pres, wp.astype(np.uint32)) and np.array_equal(multi, wm.astype(np.uint32))
    assert int((pres == 2).sum()) >= 1900 and int(multi.sum()) >= 1900
    assert engine.stats()["n_read_kmers"] == int(sum(u.size for u, _ in per))


def test_exchange_path_on_one_rank_rccl():
    """The multi-GPU path (device-side owner bucketing, all-to-all of table records, all-gathers of rare lists and
    clouds, gathered cloud view for the distance stage, mask all-reduce) executed for real with RCCL on a single
    rank — librccl loaded by cf_comm_init, ncclCommInitRank, ncclAllGather / ncclAllReduce on the context's stream —
    against the plain path, in one process and with no torch anywhere.  (N > 1 itself is covered on CPU by
    tests/test_sharded_world2.py; an N-GPU node is only available to the driver.)"""
    from centroflye_amd.sharded import ShardedRecruiter
    pk = _host.synth(seed=31, n_units=60, n_reads=200, var_len=8)
    P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
    a = ShardedRecruiter(0, force_exchange=True)
    assert a.engine.comm_info() == (0, 1)
    a.load(pk, 1)
    ra = [a.run(edge_cap=1 << 24, **P) for _ in range(2)][-1]
    ea = a.engine.edges(ra["local_edges"]); ua = a.unique_mask.copy(); ka = a.rare.copy()
    assert a.allreduce([5, 7], "max").tolist() == [5, 7]
    a.close()
    b = ShardedRecruiter(0)
    b.load(pk, 1)
    rb = b.run(edge_cap=1 << 24, **P)
    eb = b.engine.edges(rb["local_edges"]); ub = b.unique_mask.copy(); kb = b.rare.copy()
    b.close()
    srt = lambda e: e[np.lexsort((e[:, 2], e[:, 1], e[:, 0]))]
    keys = ("n_edges", "n_emissions", "n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_cloud_entries", "n_rare", "n_unique")
    assert ra["n_edges"] > 1000 and a.exchange and not b.exchange
    assert {k: ra[k] for k in keys} == {k: rb[k] for k in keys}
    assert np.array_equal(ka, kb) and np.array_equal(ua, ub) and np.array_equal(srt(ea), srt(eb))


@pytest.mark.timeout(900)
def test_p2p_rounds_execute_on_one_gpu_through_self_send_recv():
    """ncclSend / ncclRecv on real hardware: with comm_self_p2p the message a rank sends to itself goes through
    ncclSend(rank -> rank) / ncclRecv inside the same ncclGroup rounds every other pair uses, so the rounds loop of the
    all-to-all (table records) and of the variable-size all-gathers (rare lists, clouds) runs on the one GPU there is:
    (1) many small rounds (64 KiB) on a small read set, (2) the production round size (256 MB) on a read set whose table
    exchange is > 600 MB, i.e. several rounds — both against the plain path.  (N > 1 remains unmeasured: only the
    driver has a multi-GPU node.)"""
    from centroflye_amd.sharded import ShardedRecruiter
    P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)
    keys = ("n_edges", "n_emissions", "n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_cloud_entries", "n_rare", "n_unique")
    for reads, units, round_bytes, cap in ((200, 60, 65536, 1 << 24), (12000, 3600, 0, 1 << 20)):
        pk = _host.synth(seed=31, n_units=units, n_reads=reads, var_len=8)
        res = []
        for self_p2p in (0, 1):
            sr = ShardedRecruiter(0, force_exchange=bool(self_p2p))
            if self_p2p:
                sr.engine.set_param("comm_self_p2p", 1)
                if round_bytes:
                    sr.engine.set_param("comm_round_bytes", round_bytes)
            sr.load(pk, 1)
            r = sr.run(edge_cap=cap, **P)
            res.append((r, sr.rare.copy(), sr.unique_mask.copy(), sr.engine.edges_checksum() if r["local_edges"] <= cap else None, sr.exchange_bytes))
            sr.close()
        (ra, ka, ua, ca, _), (rb, kb, ub, cb, xb) = res
        assert {k: ra[k] for k in keys} == {k: rb[k] for k in keys}, reads
        assert np.array_equal(ka, kb) and np.array_equal(ua, ub) and ca == cb, reads
        if not round_bytes:
            assert 16 * rb["n_distinct"] > 600e6, "the table exchange must span several 256 MB rounds"
