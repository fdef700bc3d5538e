"""Canonical, order-independent serialisations used to compare the reference, the oracle and
the HIP path.  Everything is expressed on k-mer STRINGS so that the
reference (dicts of str) and the array code (2-bit codes) hash to the same digest."""
import hashlib


def sha(lines):
    h = hashlib.sha256()
    for ln in lines:
        h.update(ln.encode())
        h.update(b"\n")
    return h.hexdigest()


def presence_digest(pairs):
    """pairs: iterable of (kmer_str, pres) -> digest over sorted 'kmer\\tpres' lines."""
    return sha(f"{k}\t{v}" for k, v in sorted(pairs))


def set_digest(kmers):
    return sha(sorted(kmers))


def clouds_digest(clouds):
    """clouds: list per read of list per unit of iterable of k-mer strings."""
    def gen():
        for r, units in enumerate(clouds):
            for u, c in enumerate(units):
                yield f"{r}\t{u}\t" + ",".join(sorted(c))
    return sha(gen())


def hist_digest(tuples):
    """tuples: iterable of (kmer_a, kmer_b, d, cnt) with cnt > 0."""
    return sha(f"{a} {b} {d} {c}" for a, b, d, c in sorted(tuples))


def hist_digest_codes(a, b, d, cnt, k, chunk=1 << 19, presorted=False):
    """hist_digest of (decode(a), decode(b), d, cnt) for k-mers given as 2-bit codes (A < C < G < T: code order IS string order, so the
    tuples sort by (a, b, d, cnt) as numbers) — the lines are built as bytes with numpy, a few million at a time: the full histogram
    of a fixture is 2.4e7 keys, which sorted() and f-strings take most of a minute for."""
    import numpy as np
    a = np.asarray(a, np.uint64); b = np.asarray(b, np.uint64); d = np.asarray(d, np.int64); cnt = np.asarray(cnt, np.int64)
    assert a.size == 0 or (int(a.max()) < 4 ** k and int(b.max()) < 4 ** k and int(d.min()) >= 0 and int(cnt.min()) >= 0)
    if presorted:      # (the caller sorted by a packed key of its own: checked, not trusted)
        order = np.arange(a.size)
        if a.size > 1:
            ok = cnt[1:] >= cnt[:-1]
            for hi_, lo_ in ((d[1:], d[:-1]), (b[1:], b[:-1]), (a[1:], a[:-1])):
                ok = (hi_ > lo_) | ((hi_ == lo_) & ok)
            assert bool(ok.all()), "hist_digest_codes: presorted input is not sorted"
    else:
        order = np.lexsort((cnt, d, b, a))
    lut = np.frombuffer(b"ACGT", np.uint8)
    pow10 = 10 ** np.arange(19, dtype=np.int64)
    h = hashlib.sha256()
    for lo in range(0, a.size, chunk):
        o = order[lo:lo + chunk]
        n = o.size
        ca, cb = a[o], b[o]
        nums = []
        for val in (d[o], cnt[o]):
            nv = np.ones(n, np.int64)
            for p in range(1, 19):
                if int(val.max()) < pow10[p]:
                    break
                nv += val >= pow10[p]
            nums.append((val, nv, int(nv.max())))
        w = 2 * k + 2 + nums[0][2] + 1 + nums[1][2] + 1
        mat = np.zeros((n, w), np.uint8)      # a ' ' b ' ' [d, left-aligned in a field as wide as the chunk's longest] ' ' [cnt] '\n'; zero bytes = padding
        for j in range(k):
            sh = np.uint64(2 * (k - 1 - j))
            mat[:, j] = lut[((ca >> sh) & np.uint64(3)).astype(np.uint8)]
            mat[:, k + 1 + j] = lut[((cb >> sh) & np.uint64(3)).astype(np.uint8)]
        mat[:, k] = 32; mat[:, 2 * k + 1] = 32
        c0 = 2 * k + 2
        rows = np.arange(n)
        for (val, nv, wmax), sep in zip(nums, (32, 10)):
            for q in range(wmax):      # digit q from the left
                e = nv - 1 - q
                dig = (48 + (val // pow10[np.maximum(e, 0)]) % 10).astype(np.uint8)
                mat[:, c0 + q] = np.where(e >= 0, dig, 0)
            mat[rows, c0 + nv] = sep
            c0 += wmax + 1
        h.update(mat[mat != 0].tobytes())      # (row-major: the lines in order, the padding dropped)
    return h.hexdigest()


def edge_lines_digest(lines):
    return sha(sorted(lines))
