#!/usr/bin/env python3
"""Developer tool (GPU box): randomised differential test of read recruitment (reference rr.cpp:73-90).  Every case
draws a unit (1 .. 3 000 bases, lengths around the multiples of 64 — the bit-vector blocks — drawn on purpose), a batch of reads (empty,
shorter than the unit, up to 40 000 bases; random sequence, copies of the unit or of its reverse complement under 0 - 25 % edits, N calls and
lower-case letters) and a threshold (0, small, 350, no limit), and compares the device's forward and reverse-complement distances with the
reference's own edlib (oracle/_ref/librr_ref.so, built by oracle/ref/Makefile) and with the plain-C restatement.
usage: tools/fuzz_rr.py [cases] [--seed S] [--seconds T] [--out gpurun_out/fuzz_rr.json]"""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from centroflye_amd.engine import Engine
from oracle import rr


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
seed, budget = arg("--seed", 1), arg("--seconds", 10 ** 9, float)
out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_rr.json"), str)
rng = random.Random(seed)
have_ref = rr.ref_distance(b"ACGT", b"ACGT", 1) is not None


def rand_seq(n, alphabet=b"ACGT"):
    return bytes(rng.choice(alphabet) for _ in range(n))


def noisy(s, rate):
    o = bytearray()
    for c in s:
        x = rng.random()
        if x < rate / 3:
            continue
        o.append(rng.choice(b"ACGT") if x < 2 * rate / 3 else c)
        if x > 1 - rate / 3:
            o.append(rng.choice(b"ACGT"))
    return bytes(o)


recs, t_start = [], time.time()
lib = None
if os.environ.get("CF_LIB"):
    from centroflye_amd import _lib
    lib = _lib.load(os.environ["CF_LIB"])
with Engine(0, lib) as e:
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        m = rng.choice([1, 2, 31, 63, 64, 65, 127, 128, 129, 171, 255, 256, 257, 500, 1023, 1024, 1025, 2055, 2055, 3000, rng.randint(1, 3000)])
        if os.environ.get("CF_FUZZ_SMALL"):
            m = min(m, 200)
        unit = rand_seq(m)
        rcu = rr.revcomp(unit)
        reads = []
        for _ in range(rng.choice([1, 5, 20, 60])):
            kind = rng.random()
            left, right = rand_seq(rng.choice([0, 0, 7, 300, 3000])), rand_seq(rng.choice([0, 0, 9, 500, 3000]), rng.choice([b"ACGT", b"ACGTN", b"ACGTacgtN"]))
            if kind < 0.15:
                core = b""
            elif kind < 0.3:
                core = rand_seq(rng.randint(0, 2 * m + 5))
            else:
                base = (unit, rcu)[rng.random() < 0.4]
                core = noisy(base * rng.choice([1, 1, 2, 5]) if rng.random() < 0.8 else base[rng.randint(0, m - 1):], rng.choice([0.0, 0.02, 0.1, 0.17, 0.25]))
            r = left + core + right
            if os.environ.get("CF_FUZZ_SMALL"):
                r = r[:600]
            reads.append(r[:40000])
        thr = rng.choice([0, 1, 5, 40, 350, 350, -1, m, m // 6])
        flat = np.frombuffer(b"".join(reads), np.uint8) if reads else np.zeros(0, np.uint8)
        off = np.concatenate(([0], np.cumsum([len(r) for r in reads]))).astype(np.int64)
        fwd, rc = e.rr_distances(unit, flat, off, thr)
        want_f = [rr.distance(unit, r, thr) for r in reads]
        want_r = [rr.distance(rcu, r, thr) for r in reads]
        ok = fwd.tolist() == want_f and rc.tolist() == want_r
        ref_ok = None
        if have_ref:      # (the reference's edlib on the same pairs: pins the restatement on every case too)
            # (not for EMPTY reads: edlib's answer for an empty target is whatever its result struct held — 0 once, -585602796 another time in
            # the first run of this tool; the restatement and the device say "all of the unit inserted": m, or -1 above the threshold)
            ref_ok = all(not r or (a == rr.ref_distance(unit, r, thr) and b == rr.ref_distance(rcu, r, thr)) for r, a, b in zip(reads, want_f, want_r))
        rec = dict(case=i, unit_len=m, reads=len(reads), bases=int(off[-1]), threshold=thr, identical=bool(ok), oracle_equals_edlib=ref_ok,
                   within=int(sum(1 for a in want_f if a != -1) + sum(1 for a in want_r if a != -1)))
        recs.append(rec)
        if not ok or ref_ok is False:
            bad_i = next((j for j in range(len(reads)) if fwd[j] != want_f[j] or rc[j] != want_r[j]), -1)
            rec["first_difference"] = dict(read=bad_i, got=[int(fwd[bad_i]), int(rc[bad_i])], want=[want_f[bad_i], want_r[bad_i]], read_len=len(reads[bad_i])) if bad_i >= 0 else None
            print("DIFFERENCE:", json.dumps(rec), flush=True)
bad = [r for r in recs if not r["identical"] or r["oracle_equals_edlib"] is False]
summary = dict(seed=seed, cases=len(recs), identical=sum(1 for r in recs if r["identical"]), different=len(bad), edlib_checked=have_ref, reads=sum(r["reads"] for r in recs),
               bases=sum(r["bases"] for r in recs), distances_within_threshold=sum(r["within"] for r in recs), seconds=round(time.time() - t_start, 1))
json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
print(json.dumps(summary))
sys.exit(1 if bad else 0)
