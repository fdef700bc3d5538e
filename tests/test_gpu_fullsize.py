"""Full-size parity on a real MI355X against the plain-C oracle (oracle/c, OpenMP over the host's cores):
  * BASELINE configs[1] — 50 000 reads (~1 Gb): A1 + A2 on the GPU vs the CPU path, bit-exact: every counter, the whole
    (k-mer, pres, multi) table through an order-independent checksum, the rare set element by element;
  * BASELINE configs[0] — 1 000 reads: the whole stage 2 (A1-A6, ~3e9 pair emissions) vs the CPU path: counters, rare set,
    clouds, every selected edge (checksum + count), the unique mask;
  * BASELINE configs[2] — the same 50 000 reads: the distance stage (A5 + A6, 1.5e11 pair emissions) vs the CPU path by
    first-k-mer partition (a % 64 == p for two p: dist_cnt[d][a] is a's own dict, so a partition is an independent
    piece of the same result): emission count, edge count, edge checksum (device-side and on the copied rows), the
    unique bits of the partition; the full 64/64 CPU run is committed under profiles/r03_full_parity.json and asserted
    against a full GPU launch here and by bench.py every run;
  * placement (A4 + A8/A9) of 3 000 reads vs the C restatement of the greedy loop: identical lines; and of the 50 000
    reads of configs[2] vs the same C placer (the arg-max scan threaded).
  * BASELINE configs[3] at its own size — 500 000 reads (~10 Gb) resident on one GPU: A1, A2, A3 of all reads and ONE WHOLE rank of 8
    (a % 8 == 3, through the exchange path and the gathered view) against the committed oracle record
    profiles/r05_parity_500k_rank3.json (tools/parity_record.py); 200 000 reads the same way (profiles/r05_parity_200k.json);
  * two more workload families at size: point substitutions (var_len 1, SURVEY 8(d)'s literal model) on the bench's 50 000 reads, and
    reads of up to ~195 units with more than 2^24 rare k-mers.
  * round 6 — BASELINE configs[4]'s SHAPE (a 1 500-unit array at coverage 32 read by 100-kb reads: ~47 units per read, 60 800 pair emissions per first
    k-mer): A1-A3 and EVERY first k-mer against committed oracle records (profiles/r06_parity_cenx_varlen{8,1}.json), and both command lines + the
    polisher export end to end on a report of that shape against the CPU oracles (tools/cenx_cli_e2e.py).
The oracle side is pinned on CPU (tests/test_oracle_golden.py).  Reference: distance_based_kmer_recruitment.py:39-149,
read_placer.py:42-94."""
import os

import numpy as np
import pytest

from centroflye_amd import _host
from centroflye_amd.engine import Engine
from oracle import cport

pytestmark = pytest.mark.gpu
P = dict(k=19, max_nonuniq=3, lo=10, hi=32, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8)


def synth(reads, seed):
    return _host.synth(n_reads=reads, seed=seed, n_units=max(24, int(round(0.3 * reads))), var_len=8)


@pytest.fixture(scope="module")
def engine():
    e = Engine(0)
    yield e
    e.close()


def _record(name):
    import json
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)) as f:
        return json.load(f)


def _against_record(name, out_name, **kw):
    """The GPU against a COMMITTED oracle record (tools/parity_record.py ran oracle/c/cf_oracle_mt.c once on the GPU box's 256 host threads;
    the record holds the workload's generator parameters, every counter and the checksums of the A1 table, the rare set, the cloud CSR and
    of one first-k-mer partition): the same seeded reads, device-side checksums of every element (cf_checksum, cf_edges_checksum)."""
    import json
    import bigparity
    rec = _record(name)
    pk = bigparity.synth_workload(rec["workload"])
    assert pk.n_reads == rec["reads"] and pk.n_bases == rec["n_bases"]
    e = Engine(0)
    try:
        res = bigparity.check_record(e, pk, rec, **kw)
    finally:
        e.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", out_name), "w") as f:
        json.dump(res, f, indent=1)
    assert res["identical"], res["checks"]
    return rec, res


@pytest.mark.timeout(900)
def test_config3_500k_reads_count_clouds_and_one_whole_rank_of_8_vs_committed_oracle():
    """BASELINE configs[3] at its own size: 500 000 reads (9.96 Gb) resident on the one GPU; A1 (1.27e9 table entries), A2 (4.46e7 rare
    k-mers), A3 (6.5e8 cloud entries) of ALL reads, then ONE WHOLE rank of 8 — first k-mers a % 8 == 3: 6.3e10 pair emissions, 2.9e9
    selected edges — run the way a rank runs it: A1 on the rank's read shard, table exchange, rare-list and cloud gathers through a
    one-rank RCCL communicator with the message to itself going through ncclSend / ncclRecv, the distance stage over the gathered view in
    the table layout every rank of 8 takes (cf_tab_region26).  Oracle side: profiles/r05_parity_500k_rank3.json (561 s of A1-A3 + 650 s for
    the 16 sub-partitions a % 128 == 3 + 8 j on 256 host threads; the GPU equalled it when the record was taken)."""
    rec, res = _against_record("r05_parity_500k_rank3.json", "parity_500k_rank3.json", through_exchange=True)
    assert rec["reads"] == 500000 and rec["n_rare"] > (1 << 25) and rec["partition"]["n_parts"] == 8 and rec["partition"]["n_emissions"] > 6e10
    assert res["got"]["gathered_cloud_entries"] == rec["n_cloud_entries"]


@pytest.mark.timeout(600)
def test_config3_share_200k_reads_count_clouds_and_one_distance_partition_vs_committed_oracle():
    """200 000 reads (3.98 Gb): three bucket passes of the counting sort, 18 read-id bits, 2.2e7 rare k-mers (the 6-byte table slots with a
    7-bit distance field); oracle side: profiles/r05_parity_200k.json (round 4 ran the oracle inside the test: 224 s of every GPU suite)."""
    rec, res = _against_record("r05_parity_200k.json", "parity_200k.json")
    assert rec["n_rare"] > (1 << 24) and res["got"]["n_emissions_partition"] > 3e9      # (ranks beyond 24 bits: not the bench's table layout)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("what,record", [("point substitutions (var_len 1), 50 000 reads", "r05_parity_50k_varlen1.json"),
                                         ("long reads (> 128 units), 2^24+ rare k-mers", "r05_parity_long_reads.json"),
                                         ("cenX-shaped: 1 500-unit array, coverage 32, reads of mean 100 kb, var_len 8", "r06_parity_cenx_varlen8.json"),
                                         ("cenX-shaped, var_len 1", "r06_parity_cenx_varlen1.json")])
def test_other_workload_families_at_size_vs_committed_oracle(what, record):
    """Round 4's at-size tests all used one generator family (var_len 8, reads of ~10 units).  Two more: SURVEY 8(d)'s LITERAL model —
    copy-specific variants are point substitutions as simulate_tandem_repeat.py:15-30 makes them, so few k-mers are copy-specific, the
    clouds are small and E per base collapses (what a real HOR array looks like) — the bench's own 50 000 reads, A1-A3 and EVERY first
    k-mer (1.07e11 pair emissions; the record bench.py's workload_b asserts too); and 34 000 reads of ~40 and up to ~195 units (2.7 Gb) with
    more than 2^24 rare k-mers: distances up to 150 next to ranks beyond 24 bits, the table layout that streams rank and unit index apart,
    one first-k-mer partition of 256 (5.2e9 pair emissions, 1.3e8 edges).  Oracle side: committed records (tools/parity_record.py; the
    oracle ran inside this test until round 5: 210 s of every GPU suite)."""
    rec, res = _against_record(record, "parity_" + record[len("r05_parity_"):])
    if "cenx" in record:
        # Round 6 — BASELINE configs[4]'s SHAPE (README.md:59-75, run_all_cenX.sh:17-22 of the reference; the real reads are not available):
        # ~47 units per read and up to ~170, so every distance up to max_distance = 150 occurs and a first k-mer has ~60 000 pair emissions
        # (the default workload: ~20 000): A1-A3 whole and EVERY first k-mer against the record (363 s of the build container's 8 cores).
        import bigparity
        pk = bigparity.synth_workload(rec["workload"])      # (kept alive: units() hands out views of its arrays)
        up = pk.units(1)[0]
        assert 40 < float(np.diff(up).mean()) < 60 and int(np.diff(up).max()) > 150 and rec["partition"]["n_parts"] == 1
        assert res["got"]["n_emissions_partition"] == rec["partition"]["n_emissions"] > (3e10 if rec["workload"]["var_len"] == 8 else 6e9)
    elif rec["workload"]["var_len"] == 1:
        assert rec["n_bases"] > 9e8 and res["got"]["n_emissions_partition"] > 1e11 and rec["partition"]["n_parts"] == 1
    else:
        import bigparity
        pk = bigparity.synth_workload(rec["workload"])      # (kept alive: units() hands out views of its arrays)
        up = pk.units(1)[0]
        assert int(np.diff(up).max()) > 128 and rec["n_rare"] > (1 << 24) and res["got"]["n_emissions_partition"] > 1e9


@pytest.mark.timeout(900)
def test_both_command_lines_and_the_polisher_export_end_to_end_on_cenx_shaped_reads(tmp_path):
    """tools/cenx_cli_e2e.py: scripts/distance_based_kmer_recruitment.py -> scripts/read_placer.py -> scripts/eltr_polisher.py on a report
    of BASELINE configs[4]'s shape (194 MB of NCRF text), each output against its CPU oracle: the unique k-mers against the committed
    record of the OpenMP oracle over every first k-mer, read_positions.csv against the C placer, every exported FASTA against
    oracle/polisher.py.  The wall times go to gpurun_out/ (profiles/r06_cenx_cli_e2e_varlen8.json is a committed run of this tool)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out", "cenx_cli_e2e_varlen8.json")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "cenx_cli_e2e.py"), "--var-len", "8", "--out", out], capture_output=True, text=True, timeout=850)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    with open(out) as f:
        res = json.load(f)
    assert res["identical"] and all(res["checks"].values()) and res["placed"] > 900 and res["positions_exported"] > 1000 and res["n_unique_kmers"] > 400000


@pytest.mark.timeout(900)
@pytest.mark.parametrize("other", [dict(k=31), dict(k=11, lo=20, hi=200, max_d=20), dict(rel_threshold=0.6, min_cov=2, max_d=40),
                                   dict(max_nonuniq=0, lo=6, hi=20, min_d=3, max_d=300, min_cov=6)])
def test_other_parameters_at_3k_reads_vs_cpu(other):
    """Parameter sets other than the benchmark's at a size with real table pressure (tools/param_sweep_check.py runs them at 20 000 reads:
    profiles/r04_param_sweep.json): the 2-bit code's longest and a short k, the threshold that takes the double division, a low minimum
    coverage, another rare window, distances from 3 to 300 — A1 table, A2 rare set, A3 CSR and one first-k-mer partition of A5/A6 against
    the OpenMP oracle."""
    import bigparity
    pk = synth(3000, 3)
    base = dict(bigparity.P)
    e = Engine(0)
    try:
        bigparity.P.update(other)
        rec = bigparity.check(e, pk, 1, 4)
    finally:
        bigparity.P.clear(); bigparity.P.update(base)
        e.close()
    assert rec["identical"], (other, rec["checks"])
    assert rec["n_emissions_partition"] > 1e8 and rec["n_edges_partition"] > 1000


@pytest.mark.timeout(900)
def test_config1_50k_reads_count_and_rare_filter_vs_cpu(engine):
    pk = synth(50000, 2)                      # the bench workload itself
    assert pk.n_bases > 9e8
    up, us, ue, _ = pk.units(1)
    engine.load(pk, 1)
    engine.count_kmers(P["k"])
    keys, pres, multi = engine.table(sort=False)
    got_tchk = cport.table_checksum(keys, pres, multi)
    n_table = keys.size
    del keys, pres, multi
    #  the device-side checksums the 200 000- / 500 000-read tests rely on (cf_checksum.hip), element-wise against the
    # host's mixes of the COPIED rows: the A1 table here, the rare set and the cloud CSR below
    assert engine.checksum("table") == (got_tchk, n_table)
    n_rare = engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
    st = engine.stats()
    rare = engine.kmers()
    assert engine.checksum("kmers") == (cport.rare_checksum(rare), rare.size)
    n_ce = engine.build_clouds()
    cp_, ent_ = engine.clouds()
    assert engine.checksum("clouds") == (cport.cloud_checksum(cp_, ent_), n_ce)
    del cp_, ent_
    c, a = cport.stage2(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0, stop_after=1, want_arrays=True)
    assert (st["n_bases"], st["n_windows"], st["n_read_kmers"]) == (c["n_bases"], c["n_windows"], c["n_read_kmers"])
    assert (st["n_distinct"], n_table, st["n_kept"], n_rare) == (c["n_distinct"], c["n_distinct"], c["n_kept"], c["n_rare"])
    assert got_tchk == c["table_checksum"]
    assert cport.rare_checksum(rare) == c["rare_checksum"] and np.array_equal(rare, a["rare"])
    assert n_rare > 5_000_000 and c["n_distinct"] > 100_000_000


def _full_parity():
    import json
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_full_parity.json")) as f:
        return json.load(f)


@pytest.mark.timeout(1500)
def test_config2_50k_reads_distance_partitions_vs_cpu(engine):
    pk = synth(50000, 2)                      # the bench workload itself
    up, us, ue, _ = pk.units(1)
    engine.load(pk, 1)
    engine.count_kmers(P["k"])
    n_rare = engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
    n_ce = engine.build_clouds()
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], threads=0) as st:
        c = st.counters
        assert (n_rare, n_ce) == (c["n_rare"], c["n_cloud_entries"])
        a = st.arrays()
        assert np.array_equal(engine.kmers(), a["rare"])
        cp, ent = engine.clouds()
        assert np.array_equal(cp, a["cloud_ptr"]) and np.array_equal(ent, a["entries"])
        del cp, ent, a
        n_parts = 64
        for part in (5, 42):
            uq = np.zeros(n_rare, np.uint8)
            w = st.dist_part(part, n_parts, 0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], threads=0, unique=uq)
            assert w["n_emissions"] > 2_000_000_000 and w["n_edges"] > 30_000_000
            engine.reset_unique()
            ne = engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], part, n_parts, edge_cap=w["n_edges"] + 16)
            s2 = engine.stats()
            assert (ne, s2["n_emissions"], s2["n_unique"]) == (w["n_edges"], w["n_emissions"], w["n_unique"]), part
            assert engine.edges_checksum() == w["edge_checksum"], part
            rows = engine.edges(ne)
            assert cport.edge_checksum(rows) == w["edge_checksum"] and bool(np.all(rows[:, 1] % n_parts == part)), part
            del rows
            assert np.array_equal(engine.unique_mask(), uq.astype(bool)), part
    # the whole launch (all first k-mers, every selected edge stored) against the committed result of the 64 / 64 CPU run
    fp = _full_parity()
    assert fp["workload"]["reads"] == 50000 and fp["workload"]["seed"] == 2 and fp["n_bases"] == pk.n_bases
    engine.reset_unique()
    ne = engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=fp["n_edges"] + 16)
    s2 = engine.stats()
    assert (ne, s2["n_emissions"], s2["n_unique"]) == (fp["n_edges"], fp["n_emissions"], fp["n_unique"])
    assert engine.edges_checksum() == fp["edge_checksum"]
    assert cport.rare_checksum(engine.kmers()[engine.unique_mask()]) == fp["unique_kmers_checksum"]


@pytest.mark.timeout(1500)
def test_config2_50k_reads_placement_vs_c_placer(engine):
    from conftest import lines_from_placement
    pk = synth(50000, 2)
    engine.load(pk, 1)
    engine.count_kmers(P["k"])
    engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
    engine.build_clouds()
    engine.reset_unique()
    engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=0)
    gk = engine.kmers()[engine.unique_mask()]
    engine.set_kmers(gk, P["k"])
    engine.build_clouds()
    engine.filter_clouds(2)
    cp, ent = engine.clouds()
    up, _, _, _ = pk.units(1)
    cls = pk.classify(50000)
    rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
    got = engine.place_reads(cls, rank, 2, 2, 10, 3)
    want = cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)
    gl = lines_from_placement(pk.ids, *[x.tolist() for x in got])
    wl = lines_from_placement(pk.ids, *[x.tolist() for x in want])
    assert sum(1 for x in gl if not x.endswith("None")) > 45000
    assert gl == wl
    # the same with the third level of the arg-max that read sets of more than 131 072 reads get: 782 blocks in 98 groups of 8
    engine.set_param("place_l3", 1); engine.set_param("place_l3_shift", 3)
    try:
        again = engine.place_reads(cls, rank, 2, 2, 10, 3)
    finally:
        engine.set_param("place_l3", 0); engine.set_param("place_l3_shift", 0)
    assert all(np.array_equal(a, b) for a, b in zip(again, got))


@pytest.mark.timeout(900)
def test_config0_1k_reads_full_stage2_vs_cpu(engine):
    pk = synth(1000, 1)
    up, us, ue, _ = pk.units(1)
    c, a = cport.stage2(pk.bases, pk.read_off, up, us, ue, P["k"], P["max_nonuniq"], P["lo"], P["hi"], 0, 2 ** 62, P["min_d"], P["max_d"],
                        P["min_cov"], P["rel_threshold"], threads=0, want_arrays=True)
    assert c["n_emissions"] > 2_000_000_000 and c["n_edges"] > 10_000_000
    engine.load(pk, 1)
    engine.count_kmers(P["k"])
    assert engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"]) == c["n_rare"]
    assert np.array_equal(engine.kmers(), a["rare"])
    assert engine.build_clouds() == c["n_cloud_entries"]
    cp, ent = engine.clouds()
    assert np.array_equal(cp, a["cloud_ptr"]) and np.array_equal(ent, a["entries"])
    ne = engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=c["n_edges"])
    st = engine.stats()
    assert (ne, st["n_emissions"], st["n_unique"]) == (c["n_edges"], c["n_emissions"], c["n_unique"])
    assert (st["n_windows"], st["n_read_kmers"], st["n_distinct"], st["n_kept"]) == (c["n_windows"], c["n_read_kmers"], c["n_distinct"], c["n_kept"])
    assert cport.edge_checksum(engine.edges(ne)) == c["edge_checksum"]
    assert np.array_equal(engine.unique_mask(), a["unique"])
    # every layout of the (b, d) table on the same clouds: 6-byte slots with a 7-bit distance field, the region layout
    # (2 and 8 regions), the 8-byte slots, one 1 024-thread workgroup per CU, the filter's in-scan evaluation — the same edges
    # and unique set
    for knobs in ({"dist_int_thr": 0}, {"dist_dbits": 7}, {"dist_regions": 2}, {"dist_regions": 8}, {"dist_regions": 2, "dist_region_bytes": 1}, {"dist_wide": 1}, {"dist_wgs": 1, "dist_block": 1024}, {"dist_hot_cap": 5}):
        try:
            for name, v in knobs.items():
                engine.set_param(name, v)
            engine.reset_unique()
            ne2 = engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=c["n_edges"])
            assert (ne2, engine.stats()["n_emissions"]) == (c["n_edges"], c["n_emissions"]), knobs
            assert cport.edge_checksum(engine.edges(ne2)) == c["edge_checksum"], knobs
            assert np.array_equal(engine.unique_mask(), a["unique"]), knobs
        finally:
            for name in knobs:
                engine.set_param(name, 1 if name == "dist_int_thr" else 0)


@pytest.mark.timeout(900)
def test_placement_3k_reads_vs_c_placer(engine):
    from conftest import lines_from_placement
    pk = synth(3000, 5)
    engine.load(pk, 1)
    engine.count_kmers(P["k"])
    engine.select_rare(P["max_nonuniq"], P["lo"], P["hi"])
    engine.build_clouds()
    engine.reset_unique()
    engine.dist_edges(0, 2 ** 62, P["min_d"], P["max_d"], P["min_cov"], P["rel_threshold"], 0, 1, edge_cap=0)
    gk = engine.kmers()[engine.unique_mask()]
    assert gk.size > 50000
    engine.set_kmers(gk, P["k"])
    engine.build_clouds()
    engine.filter_clouds(2)
    cp, ent = engine.clouds()
    up, _, _, _ = pk.units(1)
    cls = pk.classify(50000)
    assert (cls == 0).sum() >= 2 and (cls == 1).sum() > 2000
    rank = np.argsort(np.argsort(np.array(pk.ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
    want = cport.place_reads(cls, rank, up, cp, ent, gk.size, 2, 2, 10, 3)
    got = engine.place_reads(cls, rank, 2, 2, 10, 3)
    wl = lines_from_placement(pk.ids, *[x.tolist() for x in want])
    gl = lines_from_placement(pk.ids, *[x.tolist() for x in got])
    assert sum(1 for x in gl if not x.endswith("None")) > 2500
    assert gl == wl
    # the default is the round-4 path (per-read score regions, one kernel per greedy iteration).  Other shapes of it — odd grids,
    # 256-thread workgroups (a 4-wave tail), 32-word posting rows (longer posting lists continue in the CSR arrays), regions that
    # start too small (the seed of a stage and then the whole run start over with larger ones) — and the round 1-3 path (hash
    # map + seen set: two kernels per iteration, three with an event list, other chunk / grid shapes): the same placement
    # round 5: the third level of the arg-max (groups of 64-read blocks, kept lazily; by itself only for more than 131 072 reads) forced
    # onto these 47 blocks in groups of 4 and 2; regions that would need more than 2^32 slots (they give up before allocating and the
    # hash-map path takes over)
    # ("place_long_rescans": 1 000 000 keeps the region path on these shapes whatever this read set's candidate rows look like — round 5: runs
    # whose reads keep having more than four candidate rows are handed to the hash-map path; -1: handed over at the first look, at iteration 64)
    defaults = {"place_mode": 2, "place_grid": 0, "place_block": 0, "place_row_words": 0, "place_slots_per_unit": 0, "place_fused": 1, "place_chunk": 2, "place_l3": 0, "place_l3_shift": 0,
                "place_long_rescans": 1000000}
    engine.set_param("place_long_rescans", 1000000)
    for knobs in ({"place_grid": 13, "place_block": 256}, {"place_long_rescans": -1}, {"place_row_words": 32, "place_grid": 7}, {"place_row_words": 64}, {"place_slots_per_unit": 2},
                  {"place_l3": 1, "place_l3_shift": 2}, {"place_l3": 1, "place_l3_shift": 1, "place_block": 256, "place_grid": 13}, {"place_slots_per_unit": 1 << 16},
                  {"place_mode": 1}, {"place_mode": 1, "place_fused": 0}, {"place_mode": 1, "place_chunk": 7, "place_grid": 13},
                  {"place_mode": 1, "place_chunk": 64, "place_grid": 512}):
        try:
            for k, v in knobs.items():
                engine.set_param(k, v)
            again = engine.place_reads(cls, rank, 2, 2, 10, 3)
        finally:
            for k, v in defaults.items():
                engine.set_param(k, v)
        assert all(np.array_equal(a, b) for a, b in zip(again, got)), knobs
    engine.set_param("place_long_rescans", 2)
