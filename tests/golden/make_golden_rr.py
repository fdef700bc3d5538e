#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8(f) rank 4 (read recruitment, scripts/read_recruitment/rr.cpp:73-90): seeded
(unit, read, threshold) cases with the distances the REFERENCE's own code returns — vendored edlib compiled from
/root/reference by oracle/ref/Makefile into oracle/_ref/librr_ref.so (build container only).

    python tests/golden/make_golden_rr.py          # writes rr_vectors.json
    python tests/golden/make_golden_rr.py --check
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import rr  # noqa: E402


def mutate(rng, s, n_edits, alphabet=b"ACGT"):
    s = bytearray(s)
    for _ in range(n_edits):
        if not s:
            break
        i = rng.randrange(len(s))
        op = rng.random()
        if op < 0.4:
            s[i] = rng.choice(alphabet)
        elif op < 0.7:
            del s[i]
        else:
            s.insert(i, rng.choice(alphabet))
    return bytes(s)


def cases():
    rng = random.Random(20260)
    out = []
    for m in (1, 7, 63, 64, 65, 127, 128, 129, 200, 513):
        unit = bytes(rng.choice(b"ACGT") for _ in range(m))
        rc = rr.revcomp(unit)
        for variant in range(4):
            inner = (unit, rc, b"", unit)[variant]
            body = mutate(rng, inner, rng.randint(0, max(1, m // 5)), b"ACGTN" if variant == 3 else b"ACGT")
            read = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 150))) + body + bytes(rng.choice(b"ACGTacgt") for _ in range(rng.randint(0, 150)))
            out.append((unit, read, rng.choice((-1, 0, 2, m // 8, m // 3, m))))
    unit = bytes(rng.choice(b"ACGT") for _ in range(2055))     # the size of DXZ1 (33 blocks), threshold of run_read_recruitment.sh:30
    rc = rr.revcomp(unit)
    for inner, edits in ((unit, 200), (rc, 300), (unit, 500), (b"", 0), (unit[:1500], 100)):
        read = bytes(rng.choice(b"ACGT") for _ in range(1800)) + mutate(rng, inner, edits) + bytes(rng.choice(b"ACGT") for _ in range(2100))
        out.append((unit, read, 350))
    # (an EMPTY read is not a case: edlib reads uninitialised memory for a zero-length target and returns garbage;
    #  this repository defines it as "m deletions", tests/test_read_recruitment.py)
    return out


def main():
    check = "--check" in sys.argv
    vec = []
    for unit, read, k in cases():
        rc = rr.revcomp(unit)
        vec.append(dict(unit=unit.decode(), read=read.decode(), threshold=k, fwd=rr.ref_distance(unit, read, k), rc=rr.ref_distance(rc, read, k)))
    assert all(v["fwd"] is not None for v in vec), "oracle/_ref/librr_ref.so is missing: make -C oracle/ref"
    path = os.path.join(HERE, "rr_vectors.json")
    if check:
        with open(path) as f:
            same = json.load(f) == vec
        print("IDENTICAL" if same else "DIFFERENT")
        sys.exit(0 if same else 1)
    with open(path, "w") as f:
        json.dump(vec, f, indent=0)
    print(len(vec), "cases;", sum(v["fwd"] != -1 or v["rc"] != -1 for v in vec), "recruited")


if __name__ == "__main__":
    main()
