"""Kernel LOGIC on the CPU: the unmodified HIP sources compiled against the host emulator
(tests/emu) and driven through the same C ABI.  Small cases only (the emulator runs every GPU
thread as a fiber); the real parity suite is tests/test_gpu_parity.py on an MI355X."""
import os

import numpy as np
import pytest

import fixtures
import pathcheck
from centroflye_amd.engine import DeviceError, Engine
from oracle import recruit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine(emu_lib):
    e = Engine(0, emu_lib)
    yield e
    e.close()


def test_scan_and_sort(engine):
    rng = np.random.default_rng(1)
    for n in (0, 1, 63, 2048, 2049, 20011):
        v = rng.integers(0, 1000, n)
        assert np.array_equal(engine.selftest_scan(v), np.concatenate([[0], np.cumsum(v)]))
        k = rng.integers(0, 2 ** 38, n, dtype=np.uint64)
        assert np.array_equal(engine.selftest_sort(k, 38), np.sort(k))
    k = rng.integers(0, 7, 5000, dtype=np.uint64)  # heavy duplicates
    assert np.array_equal(engine.selftest_sort(k, 8), np.sort(k))


def test_placement_argmax_reduction(engine):
    pathcheck.check_argmax_selftest(engine)


def test_stage2_small(engine, report, oracle_stage2):
    # max_distance 2 keeps the emulated dist kernel to a few 10^5 emissions
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_slots", 2048)
    engine.set_param("dist_block", 128)
    pathcheck.check_stage2(engine, report("lowcov"), tup)
    assert engine.stats()["n_spilled"] == 0


@pytest.mark.parametrize("bits", [4, 8])
def test_sketch_counters_of_four_bits_and_of_eight(engine, report, oracle_stage2, bits):
    """Round 6: the counting sketch's counters have 4 bits when min_cov <= 9 (twice as many in the same LDS; an add that sees 12 or more
    takes itself back, the add that would wrap a field is seen and the first k-mer falls back to "every b marked") — same results as with
    bytes; the synthetic clouds give ONE k-mer 300 postings, i.e. pairs counted far beyond 15 and beyond 255."""
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_slots", 2048)
    engine.set_param("dist_block", 128)
    engine.set_param("dist_sketch_bits", bits)
    try:
        pathcheck.check_stage2(engine, report("lowcov"), tup)
        engine.set_param("dist_slots", 4096)
        pathcheck.check_synthetic_clouds(engine)
    finally:
        engine.set_param("dist_sketch_bits", 0)


@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_edge_output_chunks_and_holes(engine, report, oracle_stage2, chunk):
    """Workgroups reserve the edge output in chunks; a pass that does not fit the rest of a chunk continues in the next one, the
    unused rest of every workgroup's last chunk is a hole that is closed after the kernel: tiny chunks make both happen often."""
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_edge_chunk", chunk)
    try:
        pathcheck.check_stage2(engine, report("lowcov"), tup, n_parts=2, check_table=False)
    finally:
        engine.set_param("dist_edge_chunk", 0)


def test_stage2_wide_table_layout(engine, report, oracle_stage2):
    # 8-byte slots (any k-mer set size); the default above is the 6-byte layout (ranks < 2^24, counts < 2^15)
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_wide", 1)
    try:
        engine.set_param("dist_slots", 1024)
        pathcheck.check_stage2(engine, report("lowcov"), tup, check_table=False)
        engine.set_param("dist_slots", 256)   # spill path of the wide layout, every pair in the exact table (no sketch)
        engine.set_param("dist_stage", 2)
        engine.set_param("dist_sketch", 0)
        pathcheck.check_stage2(engine, report("lowcov"), tup, n_parts=2, check_table=False)
        assert engine.stats()["n_spilled"] > 0
    finally:
        engine.set_param("dist_wide", 0)
        engine.set_param("dist_stage", 2048)
        engine.set_param("dist_sketch", 1)


def test_filter_falls_back_when_its_hot_list_is_full(engine, report, oracle_stage2):
    """The filter compacts the slots that reach min_cov into an LDS list and evaluates the list; more such slots than the
    list holds are evaluated inside the bucket scan instead (never seen with the sketch at real sizes: forced here)."""
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_slots", 2048)
    engine.set_param("dist_block", 128)
    engine.set_param("dist_hot_cap", 3)
    try:
        pathcheck.check_stage2(engine, report("lowcov"), tup, check_table=False)
        engine.set_param("dist_wide", 1)
        pathcheck.check_stage2(engine, report("lowcov"), tup, check_table=False)
    finally:
        engine.set_param("dist_wide", 0)
        engine.set_param("dist_hot_cap", 0)


def test_stage2_spill_and_partition(engine, report, oracle_stage2):
    tup = oracle_stage2("lowcov", max_distance=2)
    engine.set_param("dist_slots", 256)   # forces the (b, d) table to be split by a second hash of b
    engine.set_param("dist_block", 64)
    engine.set_param("dist_stage", 3)     # with the small chunks of the edge output below: rows that do not fit the rest of a chunk are
    engine.set_param("dist_edge_chunk", 16)   # often more than the staged list holds -> the marked-slot sweep writes them
    engine.set_param("dist_sketch", 0)    # every (b, d) pair goes to the exact table
    try:
        pathcheck.check_stage2(engine, report("lowcov"), tup, n_parts=3, check_table=False)
        assert engine.stats()["n_spilled"] > 0
    finally:
        engine.set_param("dist_sketch", 1)
        engine.set_param("dist_stage", 2048)
        engine.set_param("dist_edge_chunk", 0)


def test_postings_of_one_partition_by_sort_and_by_atomics(engine, report, oracle_stage2):
    """The postings of a partition of the first k-mers (a % n_parts == part: one rank of a multi-GPU run) are built by compacting the
    kept cloud entries per unit and sorting them on rank / n_parts; the histogram + fill passes of atomics stay behind a knob."""
    tup = oracle_stage2("lowcov", max_distance=2)
    for atomics in (0, 1):
        engine.set_param("dist_post_atomics", atomics)
        try:
            pathcheck.check_stage2(engine, report("lowcov"), tup, n_parts=3, check_table=False)
        finally:
            engine.set_param("dist_post_atomics", 0)


def test_long_posting_lists_take_the_multi_chunk_path(engine):
    engine.set_param("dist_slots", 4096)
    engine.set_param("dist_block", 128)
    pathcheck.check_synthetic_clouds(engine)                      # 300 postings of k-mer 0 > 256 per chunk
    pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=40, max_d=7, min_d=3, seed=5)


def test_max_distance_beyond_255_takes_the_16_bit_distance_layout(engine):
    """--max-distance > 255 (the reference has no limit): [b:32 | d:16 | sel:1 | cnt:15] slots; one read of 330 units, so
    distances up to 320 occur and unit indices pass 255."""
    engine.set_param("dist_slots", 4096)
    engine.set_param("dist_block", 128)
    try:
        pathcheck.check_synthetic_clouds(engine, n_reads=1, n_units=330, cloud=3, n_kmers=25, max_d=320, seed=3)
    finally:
        engine.set_param("dist_slots", 0)
        engine.set_param("dist_block", 0)


@pytest.mark.parametrize("dbits,n_units", [(5, 30), (6, 60), (7, 100)])
def test_narrow_slots_with_fewer_distance_bits(engine, dbits, n_units):
    """6-byte slots [d : DB | b : 32 - DB]: k-mer sets beyond 2^24 keep them when the reads are short enough in units;
    here the split is forced on small sets, the longest read makes distances up to n_units - 1 (just inside DB bits)."""
    engine.set_param("dist_slots", 4096)
    engine.set_param("dist_block", 128)
    engine.set_param("dist_dbits", dbits)
    try:
        pathcheck.check_synthetic_clouds(engine, n_reads=3, n_units=n_units, cloud=4, n_kmers=50, max_d=150, seed=dbits)
        with pytest.raises(DeviceError, match="dist_dbits"):      # one unit more than DB bits can tell apart
            pathcheck.check_synthetic_clouds(engine, n_reads=1, n_units=(1 << dbits) + 1, cloud=2, n_kmers=20, max_d=150, seed=1)
    finally:
        engine.set_param("dist_dbits", 0)
        engine.set_param("dist_slots", 0)
        engine.set_param("dist_block", 0)


@pytest.mark.parametrize("regions,stream_bytes", [(1, 0), (2, 0), (8, 0), (2, 1)])
def test_region_layout_of_the_six_byte_slots(engine, report, oracle_stage2, regions, stream_bytes):
    """Key [d : 8 | rank >> S : 24] in 2^S table regions (k-mer sets of 2^24 .. 2^27 ranks): forced here on small sets, whole
    stage 2 of a fixture and the synthetic clouds with long posting lists and tiny tables (chains that wrap inside a region).
    Both streams: four bytes per entry [unit index mod 64 | rank : 26] with the "64 or more" position in the item record
    (up to 2^26 ranks and 128 units per read: round 4) — the last case below has 250 units and takes the other one — and rank
    and unit index apart (round 3; stream_bytes = 1 forces it)."""
    engine.set_param("dist_regions", regions)
    engine.set_param("dist_region_bytes", stream_bytes)
    engine.set_param("dist_block", 128)
    try:
        engine.set_param("dist_slots", 2048)
        pathcheck.check_stage2(engine, report("lowcov"), oracle_stage2("lowcov", max_distance=2), check_table=False)
        engine.set_param("dist_slots", 256)
        engine.set_param("dist_sketch", 0)        # every pair in the exact table: full buckets, chains, partition splits
        pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=40, cloud=6, n_kmers=60, max_d=7, min_d=1, seed=regions)
        engine.set_param("dist_sketch", 1)
        engine.set_param("dist_slots", 4096)
        pathcheck.check_synthetic_clouds(engine, n_reads=1, n_units=250, cloud=3, n_kmers=25, max_d=240, seed=3)      # distances up to 240
        # reads of 65 .. 128 units: distances on both sides of 64, items whose "64 or more" position lies inside, before and behind them
        if (regions, stream_bytes) == (2, 0):
            pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=128, cloud=4, n_kmers=30, max_d=127, seed=13)
            pathcheck.check_synthetic_clouds(engine, n_reads=2, n_units=90, cloud=12, n_kmers=60, max_d=89, seed=23)      # long partner ranges: several items per posting
    finally:
        engine.set_param("dist_regions", 0)
        engine.set_param("dist_region_bytes", 0)
        engine.set_param("dist_sketch", 1)
        engine.set_param("dist_slots", 0)
        engine.set_param("dist_block", 0)


def test_stage3_against_reference_golden(engine, report, golden):
    from centroflye_amd import _host
    from oracle import ncrf
    name = "lowcov"
    g = golden(name)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name))
    lines = pathcheck.check_stage3(engine, pk, records, alns, lens, gk, g["stage3"], expect_lines=g["read_positions"])
    assert any(ln.endswith(" None") for ln in lines)


@pytest.mark.parametrize("knobs", [{"place_grid": 3, "place_block": 256}, {"place_row_words": 64, "place_grid": 1}, {"place_slots_per_unit": 1},
                                   {"place_mode": 1}, {"place_mode": 1, "place_fused": 0}, {"place_l3": 1}, {"place_slots_per_unit": 1 << 16},
                                   {"place_cmap_bits": 3}, {"place_cmap_bits": 5, "place_slots_per_unit": 1, "place_grid": 2}])
def test_stage3_launch_shapes_and_region_restart(engine, report, golden, knobs):
    """The greedy placement in other shapes of the round-4 path (cf_place2.hip: odd grids, 4-wave tails, wide posting rows,
    score regions that start too small: the seed of a stage, then the whole run, start over with larger ones) and on the round
    1-3 path (cf_place.hip): the reference's read_positions.csv every time."""
    from centroflye_amd import _host
    from oracle import ncrf
    name = "lowcov"
    g = golden(name)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name))
    # ({"place_l3": 1}: the third level of the arg-max on a read set of one block; {"place_slots_per_unit": 65536}: regions that would need
    # more than 2^32 slots give up BEFORE allocating — ADVICE round 4 — and the hash-map path takes over: the same lines)
    # ({"place_cmap_bits": 3}: the contig's overflow map starts with 8 slots and has to grow — round 5: a full map used to enlarge the score
    # regions instead, probing every slot at every add meanwhile; with regions that start too small on top, both kinds of restart in one run)
    defaults = {"place_mode": 2, "place_grid": 0, "place_block": 0, "place_row_words": 0, "place_slots_per_unit": 0, "place_fused": 1, "place_l3": 0, "place_cmap_bits": 0}
    # (the emulated device is made small for that case — 65 536 slots per unit of this fixture's few reads are 1 GB, which a 64 GB device
    # takes: 46 s of clearing on the host — and place_mode 3, which never falls back, must say why it gives up)
    small_device = knobs.get("place_slots_per_unit") == 1 << 16
    try:
        if small_device:
            os.environ["CFEMU_TOTAL_MB"] = "256"
            engine.set_param("place_mode", 3); engine.set_param("place_slots_per_unit", 1 << 16)
            with pytest.raises(DeviceError, match="more than the device has free"):
                pathcheck.check_stage3(engine, pk, records, alns, lens, gk, g["stage3"], expect_lines=g["read_positions"])
            engine.set_param("place_mode", 2)
        for k, v in knobs.items():
            engine.set_param(k, v)
        pathcheck.check_stage3(engine, pk, records, alns, lens, gk, g["stage3"], expect_lines=g["read_positions"])
    finally:
        os.environ.pop("CFEMU_TOTAL_MB", None)
        for k, v in defaults.items():
            engine.set_param(k, v)


def test_placement_third_level_of_the_argmax_on_several_groups(engine):
    """cf_place2's third level (the best candidate per GROUP of 64-read blocks, kept lazily: groups touched by a tail go out of date and
    are mended a few per tail; off by default: measured neutral at 500 000 reads) forced onto 130 reads = 3 blocks with groups of 1 and
    2 blocks, against the C placer line by line.  (GPU: 3 000 and 50 000 reads, tests/test_gpu_fullsize.py.)"""
    from centroflye_amd import _host
    from conftest import lines_from_placement
    from oracle import cport
    pk = _host.synth(seed=5, n_units=52, n_reads=130, var_len=8)
    up, _, _, _ = pk.units(1)
    engine.set_param("dist_block", 128); engine.set_param("dist_slots", 2048)
    try:
        us, ue = pk.units(1)[1], pk.units(1)[2]
        _, a = cport.stage2(pk.bases, pk.read_off, up, us, ue, 19, 3, 10, 32, 0, 2 ** 62, 1, 2, 4, 0.8, want_arrays=True)      # (stage 2 by the C oracle: the
        gk = a["rare"][a["unique"]]                                                                                      # emulated count + dist kernels take a minute here)
        assert gk.size > 2000
        engine.load(pk, 1)
        engine.set_kmers(gk, 19); engine.build_clouds(); engine.filter_clouds(2)
        cp, ent = engine.clouds()
        cls = pk.classify(50000)
        rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
        want = lines_from_placement(pk.ids, *[x.tolist() for x in cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)])
        assert sum(1 for x in want if not x.endswith("None")) > 90
        # (the last one: the region path gives the run up at its first look at the counters — iteration 64 of the 114 internal reads — and the
        # hash-map path starts over: round 5, reads with many candidate rows)
        for knobs in ({"place_l3": 1, "place_l3_shift": 1, "place_block": 256, "place_grid": 3, "place_long_rescans": 1000000},
                      {"place_l3": 1, "place_l3_shift": 2, "place_grid": 5, "place_long_rescans": 1000000}, {"place_long_rescans": -1}):
            try:
                for k, v in knobs.items():
                    engine.set_param(k, v)
                got = lines_from_placement(pk.ids, *[x.tolist() for x in engine.place_reads(cls, rank, 2, 2, 10, 3)])
            finally:
                for k in knobs:
                    engine.set_param(k, 2 if k == "place_long_rescans" else 0)
            assert got == want, knobs
    finally:
        engine.set_param("dist_block", 0); engine.set_param("dist_slots", 0)


@pytest.mark.parametrize("mode", [2, 3, 1])
def test_stage3_small_thresholds_on_both_paths(engine, report, golden, mode):
    """--min-inters below 4 makes nearly every score row a candidate row: the default (place_mode 2) sends such runs down the hash-map
    path, place_mode 3 keeps the per-read regions; both give the oracle's lines (with --min-cloud-kmer-freq 1 and --min-unit 1 on top)."""
    from centroflye_amd import _host
    from oracle import ncrf
    name = "lowcov"
    g = golden(name)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name))
    p3 = dict(g["stage3"], min_inters=2, min_unit=1, min_cloud_kmer_freq=1)
    engine.set_param("place_mode", mode)
    try:
        pathcheck.check_stage3(engine, pk, records, alns, lens, gk, p3)
    finally:
        engine.set_param("place_mode", 2)


def test_stage3_with_units_of_two_motifs(engine, report, golden):
    """--n-motif 2 (reference ncrf_parser.py get_motif_alignments(n=2): units of two stuck-together motifs): clouds, filter and every line
    of the placement against the oracle (whose unit split is pinned to the reference's G0 unit_cols_n2)."""
    from centroflye_amd import _host
    from oracle import ncrf
    name = "lowcov"
    g = golden(name)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name), keep_rows=True)
    p3 = dict(g["stage3"], n_motif=2, min_inters=40)
    lines = pathcheck.check_stage3(engine, pk, records, alns, lens, gk, p3)
    assert sum(1 for ln in lines if not ln.endswith(" None")) >= 5


def test_unit_kmer_occurrences_and_top_n(engine, report):
    import json
    with open(os.path.join(ROOT, "tests", "golden", "lowcov.unit_kmers.json")) as f:
        g = json.load(f)
    pathcheck.check_unit_kmers(engine, report("lowcov"), g, 30)      # by sort and reduce (records without the read id)
    for mode, bits in ((1, 3), (0, 0)):                                     # few buckets (tiles of one k-mer, tables that fill up); the atomic table
        engine.set_param("count_mode", mode); engine.set_param("count_bits", bits)
        try:
            pathcheck.check_unit_kmers(engine, report("lowcov"), g, 19)
        finally:
            engine.set_param("count_mode", 1); engine.set_param("count_bits", 0)


def test_errors(engine):
    # symbols other than upper-case ACGT are accepted (their windows have no 2-bit code and are skipped; see
    # tests/test_exotic_symbols.py) — except by the occurrence counts
    engine.load_arrays(np.frombuffer(b"ACGTNACGTACGTACGTACGTACGTACGTAC", np.uint8), [0, 31], [0, 1], [0], [31])
    engine.count_kmers(4)
    keys, pres, multi = engine.table()
    assert recruit.decode_kmer(keys[0], 4) == "ACGT" and len(keys) == 4 and not any("N" in recruit.decode_kmer(x, 4) for x in keys)
    with pytest.raises(DeviceError, match="symbols"):
        engine.count_occurrences(4)
    engine.load_arrays(np.frombuffer(b"ACGTACGTAC", np.uint8), [0, 10], [0, 1], [0], [10])
    with pytest.raises(DeviceError):
        engine.count_kmers(32)
    with pytest.raises(DeviceError, match="unit"):
        engine.load_arrays(np.frombuffer(b"ACGT", np.uint8), [0, 4], [0, 1], [0], [9])
    with pytest.raises(DeviceError, match="sorted"):
        engine.set_kmers(np.array([5, 3], np.uint64), 4)


def test_degenerate_inputs(engine):
    # no reads at all
    engine.load_arrays(np.zeros(0, np.uint8), [0], [0], [], [])
    engine.count_kmers(19)
    assert engine.select_rare(3, 1, 10) == 0
    assert engine.build_clouds() == 0
    assert engine.dist_edges(0, 10, 1, 150, 1, 0.8) == 0
    # reads shorter than k, a unit shorter than k, a read without units
    seq = b"ACGTACGTACGTACGTACGTACGTAAAC" + b"ACGTA" + b"TTGACCA"
    engine.load_arrays(np.frombuffer(seq, np.uint8), [0, 28, 33, 40], [0, 2, 2, 3], [0, 20, 33], [20, 28, 40])
    engine.count_kmers(19)
    st = engine.stats()
    codes, cnt = np.unique(recruit.encode_windows(seq[:28], 19), return_counts=True)   # only read 0 has windows
    assert st["n_windows"] == 10 and st["n_read_kmers"] == codes.size == 7
    keys, pres, multi = engine.table()
    assert np.array_equal(keys, codes) and (pres == 1).all() and np.array_equal(multi, (cnt > 1).astype(np.uint32))
    assert engine.select_rare(0, 1, 1) == int((cnt == 1).sum())    # max_nonuniq = 0 drops the repeated k-mers
    assert engine.select_rare(3, 1, 1) == 7
    assert engine.build_clouds() == 2                              # unit 0 (20 bases) has 2 windows; units 1, 2 are shorter than k
    cp, ent = engine.clouds()
    assert cp.tolist() == [0, 2, 2, 2]


def _presence_oracle(seqs, k):
    per = [np.unique(recruit.encode_windows(s, k), return_counts=True) for s in seqs]
    allk = np.concatenate([u for u, _ in per]); allm = np.concatenate([c > 1 for _, c in per])
    o = np.argsort(allk, kind="stable")
    keys, start, pres = np.unique(allk[o], return_index=True, return_counts=True)
    return keys, pres.astype(np.uint32), np.add.reduceat(allm[o].astype(np.int64), start).astype(np.uint32), int(sum(u.size for u, _ in per))


@pytest.mark.parametrize("bits,mode", [(1, 1), (4, 1), (11, 1), (0, 1), (0, 0)])
def test_count_sort_and_reduce_buckets_spanning_tiles(engine, bits, mode):
    """A1 by sort and reduce (cf_count2.hip): heavy k-mers (a short period repeated hundreds of times in every read, so
    one k-mer's records fill many reduce tiles and a read's records straddle tile borders), reads that share a k-mer
    once, twice or not at all, one bucket pass (bits 1, 4) and two (bits 11); against a numpy count — and the atomic
    table of round 1 (mode 0) against the same."""
    rng = np.random.default_rng(5)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    period = alpha[rng.integers(0, 4, 41)].tobytes()
    seqs = []
    for r in range(14):
        body = period * int(rng.integers(150, 400))
        noise = alpha[rng.integers(0, 4, int(rng.integers(200, 3000)))].tobytes()
        seqs.append(noise[: len(noise) // 2] + body + noise[len(noise) // 2:] + (period * 2 if r % 3 == 0 else b""))
    seqs.append(b"ACGTACGTAC")                                   # shorter than k: no window
    seqs.append(alpha[rng.integers(0, 4, 5000)].tobytes())      # shares nothing
    bases = np.frombuffer(b"".join(seqs), np.uint8)
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])])
    engine.load_arrays(bases, off, np.zeros(len(seqs) + 1, np.int64), [], [])
    engine.set_param("count_mode", mode); engine.set_param("count_bits", bits)
    try:
        engine.count_kmers(19)
        keys, pres, multi = engine.table()
        wk, wp, wm, n_rk = _presence_oracle(seqs, 19)
        assert np.array_equal(keys, wk) and np.array_equal(pres, wp) and np.array_equal(multi, wm)
        st = engine.stats()
        assert st["n_read_kmers"] == n_rk and st["n_windows"] == sum(max(0, len(s) - 18) for s in seqs)
        n = engine.select_rare(3, 2, 14)
        sel = (wm <= 3) & (wp >= 2) & (wp <= 14)
        assert n == int(sel.sum()) and np.array_equal(engine.kmers(), wk[sel])
        assert engine.stats()["n_distinct"] == wk.size and engine.stats()["n_kept"] == int((wm <= 3).sum())
    finally:
        engine.set_param("count_mode", 1); engine.set_param("count_bits", 0)
