"""SURVEY.md §8(f) rank 4 — read recruitment (reference scripts/read_recruitment/rr.cpp:73-90, vendored edlib in mode
HW).  Golden vectors hold the distances the reference's own code returned (tests/golden/make_golden_rr.py, built from
/root/reference into oracle/_ref by oracle/ref/Makefile).  CPU: the C oracle against them (and against oracle/_ref when
that library is present), the kernel source on the host emulator, the FASTA/FASTQ reader.  GPU: the HIP kernel through
the C ABI and the CLI."""
import gzip
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from oracle import rr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _vectors():
    with open(os.path.join(ROOT, "tests", "golden", "rr_vectors.json")) as f:
        return json.load(f)


def _pack(reads):
    off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return np.frombuffer(b"".join(reads), dtype=np.uint8), off


def _check_engine_on_vectors(engine):
    by_unit = {}
    for v in _vectors():
        by_unit.setdefault((v["unit"], v["threshold"]), []).append(v)
    for (unit, k), vs in by_unit.items():
        flat, off = _pack([v["read"].encode() for v in vs])
        fwd, rc = engine.rr_distances(unit.encode(), flat, off, k)
        assert fwd.tolist() == [v["fwd"] for v in vs]
        assert rc.tolist() == [v["rc"] for v in vs]


def test_oracle_against_reference_vectors():
    for v in _vectors():
        unit, read = v["unit"].encode(), v["read"].encode()
        assert rr.distance(unit, read, v["threshold"]) == v["fwd"]
        assert rr.distance(rr.revcomp(unit), read, v["threshold"]) == v["rc"]


def test_oracle_against_the_reference_library_when_present():
    if rr.ref_distance(b"ACGT", b"ACGT", 1) is None:
        pytest.skip("oracle/_ref/librr_ref.so not built (needs /root/reference: make -C oracle/ref)")
    rng = random.Random(5)
    for _ in range(300):
        m = rng.choice([1, 5, 63, 64, 65, 100, 128, 129, 300, 700])
        unit = bytes(rng.choice(b"ACGT") for _ in range(m))
        read = bytes(rng.choice(b"ACGTacgtN") for _ in range(rng.choice([1, 2, 10, 100, 900])))   # never empty: edlib returns garbage then
        if rng.random() < 0.7:
            cut = rng.randrange(m + 1)
            read += unit[:cut] + bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 5))) + unit[cut:]
        k = rng.choice([-1, 0, 3, m // 10, m // 4, m])
        assert rr.distance(unit, read, k) == rr.ref_distance(unit, read, k)


def test_kernel_source_on_the_host_emulator(emu_lib):
    from centroflye_amd.engine import Engine
    e = Engine(0, emu_lib)
    try:
        small = [v for v in _vectors() if len(v["unit"]) <= 200]
        for v in small[::3]:
            flat, off = _pack([v["read"].encode(), b"", v["read"].encode()[::-1]])
            unit = v["unit"].encode()
            fwd, rc = e.rr_distances(unit, flat, off, v["threshold"])
            assert (fwd[0], rc[0]) == (v["fwd"], v["rc"])
            assert fwd[1] == rr.distance(unit, b"", v["threshold"])
            assert fwd[2] == rr.distance(unit, v["read"].encode()[::-1], v["threshold"])
        with pytest.raises(Exception, match="ACGT"):
            e.rr_distances(b"ACGN", np.zeros(0, np.uint8), np.zeros(1, np.int64), 3)
        fwd, rc = e.rr_distances(b"ACGT", np.zeros(0, np.uint8), np.zeros(1, np.int64), 3)
        assert fwd.size == 0 and rc.size == 0
        # a batch of EMPTY reads has no bytes at all (round 5, tools/fuzz_rr.py: the library called that "bad arguments")
        fwd, rc = e.rr_distances(b"ACGTA", np.zeros(0, np.uint8), np.zeros(4, np.int64), 5)
        assert fwd.tolist() == [5, 5, 5] == rc.tolist() and e.rr_distances(b"ACGTA", np.zeros(0, np.uint8), np.zeros(3, np.int64), 4)[0].tolist() == [-1, -1]
        with pytest.raises(Exception, match="decrease"):
            e.rr_distances(b"ACGT", np.frombuffer(b"ACGTACGT", np.uint8), np.array([0, 6, 4, 8], np.int64), 3)
    finally:
        e.close()


def test_fasta_fastq_reader_follows_kseq(tmp_path):
    from centroflye_amd import read_recruitment as R
    fa = os.path.join(str(tmp_path), "a.fasta")
    open(fa, "w").write(">r1 some comment\nACGT\nAC\n\n>r2\tx\nGG\n>r3\n")
    assert list(R.iter_seqs(fa)) == [(b"r1", b"ACGTAC"), (b"r2", b"GG"), (b"r3", b"")]
    fq = os.path.join(str(tmp_path), "b.fq.gz")
    with gzip.open(fq, "wb") as f:
        f.write(b"@q1 c\nACGT\nTT\n+\n@@II\n>I\n@q2\nAA\n+q2\n>>\n")
    assert list(R.iter_seqs(fq)) == [(b"q1", b"ACGTTT"), (b"q2", b"AA")]


@pytest.mark.gpu
def test_gpu_distances_against_reference_vectors():
    from centroflye_amd.engine import Engine
    e = Engine(0)
    try:
        _check_engine_on_vectors(e)
    finally:
        e.close()


@pytest.mark.gpu
def test_gpu_against_oracle_on_long_reads_and_cli(tmp_path):
    from centroflye_amd.engine import Engine
    rng = random.Random(11)
    unit = bytes(rng.choice(b"ACGT") for _ in range(2055))
    rcu = rr.revcomp(unit)

    def noisy(s, rate):
        out = bytearray()
        for c in s:
            x = rng.random()
            if x < rate / 3:
                continue
            out.append(rng.choice(b"ACGT") if x < 2 * rate / 3 else c)
            if x > 1 - rate / 3:
                out.append(rng.choice(b"ACGT"))
        return bytes(out)
    reads = []
    for i in range(24):
        kind = i % 4
        core = (unit * 3, rcu * 2, b"", unit[:1000])[kind]
        reads.append(bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 3000))) + noisy(core, (0.1, 0.18, 0.0, 0.1)[kind]) + bytes(rng.choice(b"ACGTN") for _ in range(rng.randint(0, 3000))))
    flat, off = _pack(reads)
    e = Engine(0)
    try:
        fwd, rc = e.rr_distances(unit, flat, off, 350)
        assert fwd.tolist() == [rr.distance(unit, r, 350) for r in reads]
        assert rc.tolist() == [rr.distance(rcu, r, 350) for r in reads]
        assert 0 < sum((a != -1) | (b != -1) for a, b in zip(fwd, rc)) < len(reads)
        f2, _ = e.rr_distances(unit, flat, off, -1)          # no limit: never -1
        assert (f2 >= 0).all() and f2.tolist() == [rr.distance(unit, r, -1) for r in reads]
    finally:
        e.close()
    d = str(tmp_path)
    open(os.path.join(d, "unit.fasta"), "w").write(">DXZ1-like\n" + unit.decode() + "\n")
    with gzip.open(os.path.join(d, "reads.fasta.gz"), "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">read%d len=%d\n" % (i, len(r)) + r + b"\n")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rr.py"), os.path.join(d, "unit.fasta"), os.path.join(d, "reads.fasta.gz"),
                        os.path.join(d, "out.fasta"), "350"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    want = b"".join(b">read%d\n" % i + r + b"\n" for i, r in enumerate(reads) if rr.recruited(unit, [r], 350)[0])
    assert open(os.path.join(d, "out.fasta"), "rb").read() == want
