"""The oracle (numpy + C) against golden vectors captured from the REFERENCE ITSELF
(tests/golden/make_golden.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import canon
import fixtures
from oracle import cport, ncrf, placer, recruit

NAMES = list(fixtures.FIXTURES)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", NAMES)
def test_fixture_report_is_the_one_the_golden_was_made_from(name, report, golden):
    assert fixtures.sha256_file(report(name)) == golden(name)["report_sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_G0_records_units_classes(name, report, golden):
    g = golden(name)
    records, alns, lens = ncrf.parse_report(report(name))
    assert [r for r in records] == [x["r_id"] for x in g["records"]]
    for rec, x in zip(records.values(), g["records"]):
        assert (rec.strand, rec.r_len, rec.r_al_len, rec.r_st, rec.r_en) == (x["strand"], x["r_len"], x["r_al_len"], x["r_st"], x["r_en"])
        assert hashlib.sha1(rec.r_al.encode()).hexdigest() == x["r_al_sha1"]
        assert hashlib.sha1(rec.m_al.encode()).hexdigest() == x["m_al_sha1"]
        assert ncrf.unit_columns(rec, 1) == x["unit_cols"]
    for r_id, cols in g["unit_cols_n2"].items():
        assert ncrf.unit_columns(records[r_id], 2) == cols
    pre, mid, suf = ncrf.classify(records, alns, lens, g["stage3"]["prefix_threshold"])
    assert (pre, mid, suf) == (g["classify"]["prefix"], g["classify"]["internal"], g["classify"]["suffix"])
    seen = set(alns)
    assert sorted(seen - set(records)) == g["discarded"]


@pytest.mark.parametrize("name", NAMES)
def test_G1_to_G5_stage2(name, oracle_stage2, golden):
    g = golden(name)
    records, alns, lens, res, p2 = oracle_stage2(name)
    k = p2["k"]
    dec = lambda arr: [recruit.decode_kmer(c, k) for c in arr]
    assert len(res["keys"]) == g["presence"]["n"]
    assert canon.presence_digest(zip(dec(res["keys"]), (int(v) for v in res["pres"]))) == g["presence"]["digest"]
    rare_s = dec(res["rare"])
    assert len(rare_s) == g["rare"]["n"] and canon.set_digest(rare_s) == g["rare"]["digest"]
    up, cp, ent = res["unit_ptr"], res["cloud_ptr"], res["entries"]
    clouds = [[[rare_s[i] for i in ent[cp[u]:cp[u + 1]]] for u in range(up[r], up[r + 1])] for r in range(len(up) - 1)]
    assert [[len(c) for c in units] for units in clouds] == g["clouds2"]["sizes"]
    assert canon.clouds_digest(clouds) == g["clouds2"]["digest"]
    a, b, d, cnt = res["hist"]
    assert res["counters"]["E"] == g["hist"]["E"] and len(a) == g["hist"]["n_keys"]
    if name == "lowcov":   # the full (a, b, d, cnt) histogram, 2.4e7 keys: one fixture keeps the CPU suite short;
        # for the others E, the key count and every selected edge (below) pin the same histogram
        # (built from the 2-bit codes with numpy — canon.hist_digest_codes; the generic string form on a slice of it below)
        # (one sort by a packed (a, b, d) key — the rare list is ascending, so index order is string order — instead of a four-key lexsort)
        o = np.argsort((a.astype(np.uint64) << np.uint64(40)) | (b.astype(np.uint64) << np.uint64(16)) | d.astype(np.uint64), kind="stable")
        assert len(rare_s) < 1 << 24 and int(d.max()) < 1 << 16
        assert canon.hist_digest_codes(res["rare"][a[o]], res["rare"][b[o]], d[o], cnt[o], k, presorted=True) == g["hist"]["digest"]
        sl = slice(0, 50000)
        assert canon.hist_digest((rare_s[x], rare_s[y], int(z), int(w)) for x, y, z, w in zip(a[sl], b[sl], d[sl], cnt[sl])) == \
            canon.hist_digest_codes(res["rare"][a[sl]], res["rare"][b[sl]], d[sl], cnt[sl], k)
    assert res["edges"].shape[0] == g["edges"]["n"]
    assert canon.edge_lines_digest(recruit.edges_file_lines(res["rare"], res["edges"], k)) == g["edges"]["digest"]
    text = recruit.kmers_file_text(res["rare"], res["unique"], k)
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        assert text == f.read()
    assert hashlib.sha256(text.encode()).hexdigest() == g["unique_kmers"]["sha256"]


@pytest.mark.parametrize("name", NAMES)
def test_G3_G6_stage3(name, report, golden):
    g = golden(name)
    p3 = g["stage3"]
    records, alns, lens = ncrf.parse_report(report(name))
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    r3 = placer.stage3(records, alns, lens, gk, n_motif=p3["n_motif"], k_cloud=p3["k_cloud"],
                       min_cloud_kmer_freq=p3["min_cloud_kmer_freq"], min_kmer_mult=p3["min_kmer_mult"],
                       min_unit=p3["min_unit"], min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    ks = [recruit.decode_kmer(c, p3["k_cloud"]) for c in gk]
    up = r3["unit_ptr"]

    def as_clouds(cp, ent):
        return [[[ks[i] for i in ent[cp[u]:cp[u + 1]]] for u in range(up[r], up[r + 1])] for r in range(len(up) - 1)]
    assert canon.clouds_digest(as_clouds(r3["cloud_ptr"], r3["entries"])) == g["clouds3"]["digest"]
    cf = as_clouds(r3["f_cloud_ptr"], r3["f_entries"])
    assert [[len(c) for c in units] for units in cf] == g["clouds3_filtered"]["sizes"]
    assert canon.clouds_digest(cf) == g["clouds3_filtered"]["digest"]
    placed = [ln for ln in r3["lines"] if not ln.endswith(" None")]
    none = sorted(ln for ln in r3["lines"] if ln.endswith(" None"))
    assert placed == g["read_positions"]["placed"]      # byte-for-byte, in order
    assert none == g["read_positions"]["none"]          # as a sorted set (reference order is hash-seed dependent)


def test_lowcov_golden_has_a_none_tail(golden):
    assert len(golden("lowcov")["read_positions"]["none"]) > 0


def test_rare_bounds_follow_python_doubles():
    assert recruit.rare_bounds(0.9, 3.0, 32, 0.34) == (10, 32)     # 9.792 .. 32.64
    assert recruit.rare_bounds(0.9, 3.0, 10, 0.34) == (4, 10)      # 3.06 .. 10.2
    assert recruit.rare_bounds(1.0, 2.0, 10, 0.5) == (5, 10)       # exact integers are inclusive
    lo, hi = recruit.rare_bounds(0.0, 0.0, 32, 0.34)
    assert (lo, hi) == (0, 0)


@pytest.mark.parametrize("name", ["lowcov"])
def test_c_oracle_equals_numpy_oracle(name, report, oracle_stage2):
    from centroflye_amd import _host
    records, alns, lens, res, p2 = oracle_stage2(name)
    pk = _host.parse_report(report(name))
    up, us, ue, _ = pk.units(1)
    c, a = cport.stage2(pk.bases, pk.read_off, up, us, ue, p2["k"], p2["max_nonuniq"], res["counters"]["lo"], res["counters"]["hi"],
                        0, 2 ** 62, p2["min_distance"], p2["max_distance"], p2["min_coverage"], 0.8, want_arrays=True)
    assert np.array_equal(a["rare"], res["rare"])
    assert np.array_equal(a["cloud_ptr"], res["cloud_ptr"]) and np.array_equal(a["entries"], res["entries"])
    ed = a["edges"].astype(np.int64)
    ed = ed[np.lexsort((ed[:, 2], ed[:, 1], ed[:, 0]))]
    assert np.array_equal(ed, res["edges"])
    assert np.array_equal(np.flatnonzero(a["unique"]), res["unique"])
    cn = res["counters"]
    assert (c["n_windows"], c["n_read_kmers"], c["n_distinct"], c["n_emissions"]) == (cn["n_w"], cn["n_rk"], cn["n_distinct"], cn["E"])
    assert c["edge_checksum"] == cport.edge_checksum(res["edges"])
    assert c["rare_checksum"] == cport.rare_checksum(res["rare"])
    assert c["cloud_checksum"] == cport.cloud_checksum(res["cloud_ptr"], res["entries"])


@pytest.mark.parametrize("name", NAMES)
def test_openmp_c_oracle_equals_single_thread_c_oracle(name, report):
    """cf_oracle_mt.c (the checker of the full-size GPU tests and the all-core CPU baseline) == cf_oracle.c, which is
    pinned to the numpy oracle and through it to the reference goldens; also the table checksum against numpy."""
    from centroflye_amd import _host
    p2 = fixtures.stage2_params(name)
    lo, hi = recruit.rare_bounds(p2["bottom"], p2["top"], p2["coverage"], p2["kmer_survival_rate"])
    pk = _host.parse_report(report(name))
    up, us, ue, _ = pk.units(1)
    args = (pk.bases, pk.read_off, up, us, ue, p2["k"], p2["max_nonuniq"], lo, hi, 0, 2 ** 62, p2["min_distance"], min(p2["max_distance"], 6), p2["min_coverage"], 0.8)
    c1, a1 = cport.stage2(*args, want_arrays=True)
    for threads in (3, 0):
        c2, a2 = cport.stage2(*args, want_arrays=True, threads=threads)
        assert {k: v for k, v in c2.items() if k != "table_checksum"} == c1
        assert np.array_equal(a1["rare"], a2["rare"]) and np.array_equal(a1["cloud_ptr"], a2["cloud_ptr"]) and np.array_equal(a1["entries"], a2["entries"])
        srt = lambda e: e[np.lexsort((e[:, 2], e[:, 1], e[:, 0]))]
        assert np.array_equal(srt(a1["edges"]), srt(a2["edges"])) and np.array_equal(a1["unique"], a2["unique"])
    # the table checksum against an independent numpy count of (k-mer, pres, multi) over all reads
    per = [np.unique(recruit.encode_windows(pk.bases[pk.read_off[r]:pk.read_off[r + 1]].tobytes(), p2["k"]), return_counts=True) for r in range(pk.n_reads)]
    allk = np.concatenate([u for u, _ in per]); allm = np.concatenate([c > 1 for _, c in per])
    o = np.argsort(allk, kind="stable")
    keys, start, pres = np.unique(allk[o], return_index=True, return_counts=True)
    multi = np.add.reduceat(allm[o].astype(np.int64), start)
    assert c2["table_checksum"] == cport.table_checksum(keys, pres, multi)
    c3, _ = cport.stage2(*args, threads=2, stop_after=1)
    assert (c3["n_rare"], c3["rare_checksum"], c3["n_distinct"], c3["n_kept"], c3["table_checksum"]) == (c1["n_rare"], c1["rare_checksum"], c1["n_distinct"], c1["n_kept"], c2["table_checksum"])
    assert c3["n_cloud_entries"] == 0 and c3["n_emissions"] == 0
    # the partitioned form (the checker of the 50 000-read distance stage, bench.py's same-data CPU baseline): the first
    # k-mers a % n == p for every p give disjoint pieces that add up to the whole result, with any thread count
    with cport.Stage2State(pk.bases, pk.read_off, up, us, ue, p2["k"], p2["max_nonuniq"], lo, hi, threads=3) as st:
        assert {k: st.counters[k] for k in ("n_rare", "n_cloud_entries", "rare_checksum", "cloud_checksum", "table_checksum")} == \
            {k: c2[k] for k in ("n_rare", "n_cloud_entries", "rare_checksum", "cloud_checksum", "table_checksum")}
        arr = st.arrays()
        assert np.array_equal(arr["rare"], a1["rare"]) and np.array_equal(arr["cloud_ptr"], a1["cloud_ptr"]) and np.array_equal(arr["entries"], a1["entries"])
        for n_parts in (1, 3):
            uq = np.zeros(c1["n_rare"], np.uint8)
            rows, em, chk = [], 0, 0
            for part in range(n_parts):
                w = st.dist_part(part, n_parts, 0, 2 ** 62, p2["min_distance"], min(p2["max_distance"], 6), p2["min_coverage"], 0.8,
                                 threads=1 + part, unique=uq, want_edges=c1["n_edges"])
                assert np.all(w["edges"][:, 1] % n_parts == part) and cport.edge_checksum(w["edges"]) == w["edge_checksum"]
                rows.append(w["edges"]); em += w["n_emissions"]; chk = (chk + w["edge_checksum"]) % 2 ** 64
            assert (em, chk) == (c1["n_emissions"], c1["edge_checksum"])
            assert np.array_equal(srt(np.concatenate(rows)), srt(a1["edges"])) and np.array_equal(uq.astype(bool), a1["unique"])


@pytest.mark.parametrize("name", NAMES)
def test_c_placer_equals_python_placer(name, report, golden):
    """cf_oracle_place.c (the checker of the GPU placement test at thousands of reads) writes the lines oracle/placer.py
    writes, which are the reference's own read_positions.csv (test_G3_G6_stage3)."""
    g = golden(name)
    p3 = g["stage3"]
    records, alns, lens = ncrf.parse_report(report(name))
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    r3 = placer.stage3(records, alns, lens, gk, n_motif=p3["n_motif"], k_cloud=p3["k_cloud"],
                       min_cloud_kmer_freq=p3["min_cloud_kmer_freq"], min_kmer_mult=p3["min_kmer_mult"],
                       min_unit=p3["min_unit"], min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    ids = list(records)
    rank = np.argsort(np.argsort(np.array(ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
    rd, pos, s0, s1 = cport.place_reads(r3["classes"], rank, r3["unit_ptr"], r3["f_cloud_ptr"], r3["f_entries"], gk.size,
                                        p3["min_cloud_kmer_freq"], p3["min_unit"], p3["min_inters"], 3)
    from conftest import lines_from_placement
    assert lines_from_placement(ids, rd.tolist(), pos.tolist(), s0.tolist(), s1.tolist()) == r3["lines"]


@pytest.mark.parametrize("name", NAMES)
def test_unit_kmer_oracle_against_reference_golden(name, report):
    """§8(f) rank 2: occurrence counts and top-n (k = 30 and 19) of oracle/unit_kmers.py vs the reference."""
    import json
    from oracle import unit_kmers
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.unit_kmers.json")) as f:
        g = json.load(f)
    assert fixtures.sha256_file(report(name)) == g["report_sha256"]
    records, _, _ = ncrf.parse_report(report(name))
    recs = list(records.values())
    seqs = [r.r_al.replace("-", "").encode() for r in recs]
    for k in (30, 19):
        e = g["k"][str(k)]
        keys, cnt = unit_kmers.kmer_occurrences(seqs, k)
        assert keys.size == e["n_distinct"] and int(cnt.sum()) == e["total"]
        assert canon.presence_digest((recruit.decode_kmer(c, k), int(v)) for c, v in zip(keys, cnt)) == e["counts_digest"]
        n = 3 * unit_kmers.n_circular_unit_kmers(recs[0].motif, k)
        top = unit_kmers.most_frequent(keys, cnt, n)
        strs = [recruit.decode_kmer(keys[i], k) for i in top]
        assert len(strs) == e["n_top"] and canon.set_digest(strs) == e["top_digest"]
        assert [[s_, int(cnt[i])] for s_, i in zip(strs[:20], top[:20])] == e["top_head"]
