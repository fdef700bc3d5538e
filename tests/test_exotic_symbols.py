"""Reads with symbols the 2-bit k-mer code does not have — soft-masked (lower-case) stretches and N calls — through the
drop-in CLIs, against files the REFERENCE wrote from the same report (tests/golden/exotic.json, captured by
make_golden.py under two hash seeds).  Reference semantics: A1 counts the windows of the RAW row
(distance_based_kmer_recruitment.py:47-53), so a window holding such a symbol is a k-mer of its own; A3 upper-cases the row
first (read_kmer_cloud.py:25).  Here the device skips those windows in A1 (the host side-path cfh_exotic_summary counts them),
upper-cases in A3, and the files come out identical.  Fixture "exotic": none of those windows can reach an output (rare ones hold
lower-case letters).  Fixture "exotic_rare": the same base is an N in every read that covers it, so k-mers with an N ARE rare — the
reference selects them, finds them in the upper-cased units and writes them; here they travel beside the 2-bit set as strings
(kmers.KmerSet.extra), take ranks on the device through pseudo-codes, and get their cloud entries from the host."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import canon
import fixtures
from centroflye_amd import _host, distance_based_kmer_recruitment as dbkr, read_placer, session
from centroflye_amd.engine import Engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def g():
    with open(os.path.join(ROOT, "tests", "golden", "exotic.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def exotic_report(fx_dir, g):
    path = fixtures.make_report("exotic", fx_dir)
    assert fixtures.sha256_file(path) == g["report_sha256"]
    return path


def _argv2(report, outdir, p2):
    return ["--ncrf", report, "--coverage", str(p2["coverage"]), "--min-coverage", str(p2["min_coverage"]), "--outdir", outdir,
            "-k", str(p2["k"]), "--max-distance", str(p2["max_distance"]), "--min-distance", str(p2["min_distance"])]


def _argv3(report, kfile, outdir, p3):
    return ["--ncrf", report, "--genomic-kmers", kfile, "--outdir", outdir, "--n-motif", str(p3["n_motif"]),
            "--min-cloud-kmer-freq", str(p3["min_cloud_kmer_freq"]), "--min-kmer-mult", str(p3["min_kmer_mult"]),
            "--min-unit", str(p3["min_unit"]), "--min-inters", str(p3["min_inters"]), "--prefix-threshold", str(p3["prefix_threshold"])]


def _check_files(out2, out3, g):
    p2 = g["stage2"]
    with open(os.path.join(out2, f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt"), "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == g["unique_kmers"]["sha256"]
    with open(os.path.join(out2, f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt")) as f:
        elines = f.read().splitlines()
    assert len(elines) == g["edges"]["n"] and canon.edge_lines_digest(elines) == g["edges"]["digest"]
    with open(os.path.join(out3, "read_positions.csv")) as f:
        lines = f.read().splitlines()
    assert [ln for ln in lines if not ln.endswith(" None")] == g["read_positions"]["placed"]
    assert sorted(ln for ln in lines if ln.endswith(" None")) == g["read_positions"]["none"]


def test_the_report_really_holds_such_symbols_and_the_side_path_counts_them(exotic_report, g):
    pk = _host.parse_report(exotic_report)
    assert pk.non_acgt
    raw = pk.bases.tobytes()
    assert raw.count(b"N") >= 5 and sum(raw.count(c) for c in (b"a", b"c", b"g", b"t")) >= 500
    p2 = g["stage2"]
    lo, hi = dbkr.rare_window(p2["bottom"], p2["top"], p2["coverage"], p2["kmer_survival_rate"])
    ex = pk.exotic_summary(p2["k"], p2["max_nonuniq"], lo, hi)
    # an independent count of the same windows
    seen = {}
    for r in range(pk.n_reads):
        s = raw[pk.read_off[r]:pk.read_off[r + 1]]
        mine = {}
        for w in range(len(s) - p2["k"] + 1):
            x = s[w:w + p2["k"]]
            if x.strip(b"ACGT"):
                mine[x] = mine.get(x, 0) + 1
        for x, c in mine.items():
            a = seen.setdefault(x, [0, 0]); a[0] += 1; a[1] += c > 1
    assert ex["n_distinct"] == len(seen) > 100 and ex["n_read_kmers"] == sum(v[0] for v in seen.values())
    assert ex["n_kept"] == sum(v[1] <= p2["max_nonuniq"] for v in seen.values())
    assert ex["n_rare"] == sum(v[1] <= p2["max_nonuniq"] and lo <= v[0] <= hi for v in seen.values())
    assert ex["n_blocking"] == 0


def test_presence_and_rare_counts_add_up_to_the_reference(emu_lib, exotic_report, g):
    """reference's dict of kept k-mers (G1) = kept 2-bit k-mers of the device + kept k-mers of the host side-path; its rare
    set (G2) likewise."""
    pk = _host.parse_report(exotic_report)
    p2 = g["stage2"]
    lo, hi = dbkr.rare_window(p2["bottom"], p2["top"], p2["coverage"], p2["kmer_survival_rate"])
    ex = pk.exotic_summary(p2["k"], p2["max_nonuniq"], lo, hi)
    with Engine(0, emu_lib) as e:
        e.load(pk, 1)
        e.count_kmers(p2["k"])
        n_rare = e.select_rare(p2["max_nonuniq"], lo, hi)
        st = e.stats()
    assert st["n_kept"] + ex["n_kept"] == g["presence"]["n"]
    assert n_rare + ex["n_rare"] == g["rare"]["n"]


def test_rare_kmer_with_an_N_is_carried_as_a_string(emu_lib, tmp_path):
    """A k-mer with an N that IS rare (the same N-holding window in many reads) matches windows of the upper-cased units: the rare
    set carries it beside the 2-bit codes, and the clouds of the units that hold it get its rank."""
    unit = "ACGTTGCAAGGCTTAACCGGATCGATTACAGGCATCGGAT"
    lines = []
    for r in range(12):
        row = unit * 200
        row = row[:333] + "N" + row[334:]
        lines += [f"read{r} {len(row) + 50} {len(row)}bp 10-{10 + len(row)} {row}", f"{unit}+ {len(row)}bp score=9 {row}"]
    path = tmp_path / "n.ncrf"
    path.write_text("\n".join(lines) + "\n")
    pk = _host.parse_report(str(path))
    ex = pk.exotic_summary(19, 3, 10, 32)
    assert ex["n_blocking"] == 19
    assert dbkr.check_exotic_windows(pk, 19, 3, 10, 32)["n_blocking"] == 19
    strs = pk.exotic_rare(19, 3, 10, 32)
    row = (unit * 200)[:333] + "N" + (unit * 200)[334:]
    assert strs == sorted(row[333 - j:333 - j + 19] for j in range(19))
    from centroflye_amd import kmers as km, read_kmer_cloud as rkc
    kset = km.KmerSet(np.zeros(0, np.uint64), 19, strs)
    hu, hr = rkc.exotic_hits(pk, 1, kset)
    _, us, ue, _ = pk.units(1)
    want = sorted((u, kset.index(row[w:w + 19])) for r in range(pk.n_reads) for u in range(pk.units(1)[0][r], pk.units(1)[0][r + 1])
                  for w in range(333 - 18, 334) if us[u] - pk.read_off[r] <= w and w + 19 <= ue[u] - pk.read_off[r])
    assert list(zip(hu.tolist(), hr.tolist())) == want and len(want) >= 12


def test_rare_soft_masked_windows_are_members_of_the_rare_set_only(emu_lib, tmp_path):
    """The same soft-masked stretch in twelve reads: its windows are RARE k-mers of the reference's get_rare_kmers (the raw text, :47-53 / :66-82)
    and keys of its kmer_index — and of nothing else: no cloud holds them (read_kmer_cloud.py:25 upper-cases the unit), so no edge and no line
    of the k-mer file.  Round 5 (tools/fuzz_api2_vs_reference.py): the returned set left them out."""
    from centroflye_amd import read_kmer_cloud as rkc
    from centroflye_amd.ncrf_parser import NCRF_Report
    unit = "ACGTTGCAAGGCTTAACCGGATCGATTACAGGCATCGGAT"
    lines = []
    for r in range(12):
        row = unit * 200
        row = row[:333] + row[333:363].lower() + row[363:]
        lines += [f"read{r} {len(row) + 50} {len(row)}bp 10-{10 + len(row)} {row}", f"{unit}+ {len(row)}bp score=9 {row}"]
    path = tmp_path / "lower.ncrf"
    path.write_text("\n".join(lines) + "\n")
    session.reset()
    session._engine = Engine(0, emu_lib)
    try:
        rep = NCRF_Report(str(path))
        rare = dbkr.get_rare_kmers(rep, k=19, bottom=0.9, top=3.0, coverage=32, kmer_survival_rate=0.34, max_nonuniq=3, verbose=False)
        row = (unit * 200)[:333] + (unit * 200)[333:363].lower() + (unit * 200)[363:]
        want = sorted({row[w:w + 19] for w in range(333 - 18, 363) if row[w:w + 19] != row[w:w + 19].upper()})
        assert list(rare.inert) == want and len(want) == 48 and all(x in rare for x in want) and len(rare) == rare.codes.size + 48
        assert "acgttgcaaggcttaaccg" not in rare and set(rare) >= set(want)
        clouds = rkc.get_reads_kmer_clouds(rep, n=1, k=19, genomic_kmers=rare)
        assert not any(x in want for c in clouds.values() for u in c.kmers for x in u)
        dist_cnt, kmer_index = dbkr.get_kmer_dist_map(clouds, rare, 0, 2 ** 62, 1, 2, False)
        assert len(kmer_index) == len(rare) and sorted(kmer_index[x] for x in want) == list(range(rare.codes.size, rare.codes.size + 48))
        uniq, edges = dbkr.filter_dist_tuples(dist_cnt, 4)
        assert all(max(a, b) < rare.codes.size for _, a, b, _ in edges) and all(u < rare.codes.size for u in uniq)
    finally:
        session.reset()


@pytest.fixture(scope="module")
def g_rare():
    with open(os.path.join(ROOT, "tests", "golden", "exotic_rare.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def exotic_rare_report(fx_dir, g_rare):
    path = fixtures.make_report("exotic_rare", fx_dir)
    assert fixtures.sha256_file(path) == g_rare["report_sha256"]
    return path


def test_rare_kmers_with_an_N_reach_the_files_as_in_the_reference_on_emulated_kernels(emu_lib, exotic_rare_report, g_rare, tmp_path):
    """The reference's own outputs for a report whose N-holding k-mers are rare: 35 of its unique k-mers hold an N."""
    with open(os.path.join(ROOT, "tests", "golden", "exotic_rare.unique_kmers.txt")) as f:
        assert sum("N" in ln for ln in f) >= 30
    session.reset()
    session._engine = Engine(0, emu_lib)
    session._engine.set_param("dist_slots", 2048)
    session._engine.set_param("dist_block", 128)
    try:
        out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
        dbkr.main(_argv2(exotic_rare_report, out2, g_rare["stage2"]))
        kfile = os.path.join(out2, f"unique_kmers_min_edge_cov_{g_rare['stage2']['min_coverage']}.txt")
        with open(kfile) as f, open(os.path.join(ROOT, "tests", "golden", "exotic_rare.unique_kmers.txt")) as h:
            assert f.read() == h.read()
        read_placer.main(_argv3(exotic_rare_report, kfile, out3, g_rare["stage3"]))
        _check_files(out2, out3, g_rare)
    finally:
        session.reset()


@pytest.mark.gpu
def test_rare_kmers_with_an_N_reach_the_files_as_in_the_reference_on_the_gpu(exotic_rare_report, g_rare, tmp_path):
    out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "distance_based_kmer_recruitment.py")] + _argv2(exotic_rare_report, out2, g_rare["stage2"]),
                          stdout=subprocess.DEVNULL)
    kfile = os.path.join(out2, f"unique_kmers_min_edge_cov_{g_rare['stage2']['min_coverage']}.txt")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "read_placer.py")] + _argv3(exotic_rare_report, kfile, out3, g_rare["stage3"]),
                          stdout=subprocess.DEVNULL)
    _check_files(out2, out3, g_rare)


@pytest.mark.parametrize("name", ["exotic", "exotic_rare"])
def test_presence_mapping_and_rare_set_equal_the_reference_key_by_key(emu_lib, fx_dir, name):
    """G1 and G2 of the reference on reads with N calls and soft-masked stretches: the mapping k-mer -> reads holding it (keys with such
    symbols are strings of the raw text, reference :47-53) and the rare set, digests over every key."""
    from centroflye_amd.ncrf_parser import NCRF_Report
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.json")) as f:
        gg = json.load(f)
    p2 = gg["stage2"]
    session.reset()
    session._engine = Engine(0, emu_lib)
    try:
        rep = NCRF_Report(fixtures.make_report(name, fx_dir), keep_rows=False)
        freqs = dbkr.get_kmer_freqs_from_ncrf_report(rep, k=p2["k"], verbose=False, max_nonuniq=p2["max_nonuniq"])
        assert len(freqs) == gg["presence"]["n"] and canon.presence_digest(freqs.items()) == gg["presence"]["digest"]
        rare = dbkr.get_rare_kmers(rep, k=p2["k"], bottom=p2["bottom"], top=p2["top"], coverage=p2["coverage"],
                                   kmer_survival_rate=p2["kmer_survival_rate"], max_nonuniq=p2["max_nonuniq"], verbose=False)
        # (the reference's rare set also holds rare windows WITH lower-case letters: members of the returned set since round 5 —
        # tools/fuzz_api2_vs_reference.py —, they can match nothing downstream)
        assert len(rare) == gg["rare"]["n"] and canon.set_digest(list(rare)) == gg["rare"]["digest"]
        lo, hi = dbkr.rare_window(p2["bottom"], p2["top"], p2["coverage"], p2["kmer_survival_rate"])
        assert sorted(rare.inert) == sorted(s for s, v in freqs.extra.items() if lo <= v <= hi and s != s.upper())
    finally:
        session.reset()


def _sharded_counts(lib, report, g_, tmp_path):
    """the exchange path (bucketing, all-to-all, all-gathers) on ONE rank: edges and unique k-mers of the reference's files"""
    from centroflye_amd.sharded import ShardedRecruiter
    p2 = g_["stage2"]
    lo, hi = dbkr.rare_window(p2["bottom"], p2["top"], p2["coverage"], p2["kmer_survival_rate"])
    sr = ShardedRecruiter(0, lib=lib, rank=0, world=1, rendezvous=str(tmp_path / "rdv"), force_exchange=True)
    try:
        if lib is not None:
            sr.engine.set_param("dist_slots", 2048); sr.engine.set_param("dist_block", 128)
        sr.load(_host.parse_report(report), 1)
        out = sr.run(k=p2["k"], max_nonuniq=p2["max_nonuniq"], lo=lo, hi=hi, min_d=p2["min_distance"], max_d=p2["max_distance"],
                     min_cov=p2["min_coverage"], rel_threshold=0.8, edge_cap=0)
        assert out["n_edges"] == g_["edges"]["n"] and out["n_unique"] == g_["unique_kmers"]["n"] == int(sr.unique_mask.sum())
        assert len(sr.kset.extra) >= 30 and sorted(sr.kset.strings(np.flatnonzero(sr.unique_mask))) == open(
            os.path.join(ROOT, "tests", "golden", "exotic_rare.unique_kmers.txt")).read().split()
    finally:
        sr.close()


def test_sharded_entry_point_carries_the_rare_N_kmers_on_emulated_kernels(emu_lib, exotic_rare_report, g_rare, tmp_path):
    _sharded_counts(emu_lib, exotic_rare_report, g_rare, tmp_path)


@pytest.mark.gpu
def test_sharded_entry_point_carries_the_rare_N_kmers_on_the_gpu(exotic_rare_report, g_rare, tmp_path):
    _sharded_counts(None, exotic_rare_report, g_rare, tmp_path)


def test_cli_files_equal_the_reference_on_emulated_kernels(emu_lib, exotic_report, g, tmp_path):
    session.reset()
    session._engine = Engine(0, emu_lib)
    session._engine.set_param("dist_slots", 2048)
    session._engine.set_param("dist_block", 128)
    try:
        out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
        dbkr.main(_argv2(exotic_report, out2, g["stage2"]))
        kfile = os.path.join(out2, f"unique_kmers_min_edge_cov_{g['stage2']['min_coverage']}.txt")
        read_placer.main(_argv3(exotic_report, kfile, out3, g["stage3"]))
        _check_files(out2, out3, g)
    finally:
        session.reset()


@pytest.mark.gpu
def test_cli_files_equal_the_reference_on_the_gpu(exotic_report, g, tmp_path):
    out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "distance_based_kmer_recruitment.py")] + _argv2(exotic_report, out2, g["stage2"]),
                          stdout=subprocess.DEVNULL)
    kfile = os.path.join(out2, f"unique_kmers_min_edge_cov_{g['stage2']['min_coverage']}.txt")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "read_placer.py")] + _argv3(exotic_report, kfile, out3, g["stage3"]),
                          stdout=subprocess.DEVNULL)
    _check_files(out2, out3, g)
