"""k-mer strings <-> 2-bit codes (A=0 C=1 G=2 T=3, first base most significant) and the
array-backed containers the drop-in modules hand out in place of the reference's dict / set of
Python strings (they keep the same mapping / set protocol for small inputs)."""
from collections.abc import Mapping, Set

import numpy as np

_LUT = np.full(256, 255, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i
_ALPHA = np.frombuffer(b"ACGT", np.uint8)


def encode(strings, k):
    """list of k-long ACGT strings -> uint64 codes; raises on anything else."""
    n = len(strings)
    if n == 0:
        return np.zeros(0, np.uint64)
    raw = np.frombuffer("".join(strings).encode(), np.uint8)
    if raw.size != n * k:
        raise ValueError(f"every k-mer must have length {k}")
    two = _LUT[raw].reshape(n, k)
    if (two == 255).any():
        raise ValueError("k-mers must be upper-case ACGT")
    w = (np.uint64(1) << (np.uint64(2) * np.arange(k - 1, -1, -1, dtype=np.uint64)))
    return (two.astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)


def try_encode(s, k):
    if not isinstance(s, str) or len(s) != k:
        return None
    v = 0
    for ch in s:
        i = "ACGT".find(ch)
        if i < 0:
            return None
        v = (v << 2) | i
    return v


def decode(codes, k):
    codes = np.asarray(codes, np.uint64)
    shifts = np.uint64(2) * np.arange(k - 1, -1, -1, dtype=np.uint64)
    chars = _ALPHA[((codes[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.intp)]
    return [row.tobytes().decode() for row in chars]


def decode_one(code, k):
    return decode(np.array([code], np.uint64), k)[0]


class KmerSet(Set):
    """Sorted unique k-mer codes behaving like the reference's ``set`` of k-mer strings."""

    def __init__(self, codes, k):
        self.codes = np.ascontiguousarray(codes, np.uint64)
        self.k = int(k)

    @classmethod
    def _from_iterable(cls, it):   # results of set algebra fall back to plain sets
        return set(it)

    def __len__(self):
        return int(self.codes.size)

    def __contains__(self, s):
        c = try_encode(s, self.k)
        if c is None:
            return False
        i = int(np.searchsorted(self.codes, np.uint64(c)))
        return i < self.codes.size and int(self.codes[i]) == c

    def __iter__(self):
        return iter(decode(self.codes, self.k))

    def index(self, s):
        c = try_encode(s, self.k)
        i = int(np.searchsorted(self.codes, np.uint64(c))) if c is not None else -1
        if c is None or i >= self.codes.size or int(self.codes[i]) != c:
            raise KeyError(s)
        return i


class KmerIndex(Mapping):
    """kmer string -> index (= rank in ascending order); the reference's ``kmer_index`` dict."""

    def __init__(self, kset):
        self.kset = kset

    def __len__(self):
        return len(self.kset)

    def __iter__(self):
        return iter(self.kset)

    def __getitem__(self, s):
        return self.kset.index(s)


class KmerFreqs(Mapping):
    """kmer string -> number of reads containing it (the reference's ``all_kmers`` dict)."""

    def __init__(self, codes, pres, k):
        self.codes = np.ascontiguousarray(codes, np.uint64)
        self.pres = np.ascontiguousarray(pres, np.int64)
        self.k = int(k)

    def __len__(self):
        return int(self.codes.size)

    def __iter__(self):
        return iter(decode(self.codes, self.k))

    def __getitem__(self, s):
        c = try_encode(s, self.k)
        i = int(np.searchsorted(self.codes, np.uint64(c))) if c is not None else -1
        if c is None or i >= self.codes.size or int(self.codes[i]) != c:
            raise KeyError(s)
        return int(self.pres[i])

    def items(self):
        return zip(decode(self.codes, self.k), (int(v) for v in self.pres))


def as_kmer_set(kmers, k):
    """Accept a KmerSet, an array of codes, or any iterable of k-mer strings."""
    if isinstance(kmers, KmerSet):
        if kmers.k != k:
            raise ValueError("k-mer length mismatch")
        return kmers
    if isinstance(kmers, np.ndarray):
        return KmerSet(np.unique(kmers.astype(np.uint64)), k)
    if kmers is None:
        raise TypeError("a k-mer set is required (the reference would fail on `kmer in None`)")
    strs = [s for s in kmers if try_encode(s, k) is not None]  # other strings can never match an ACGT window
    return KmerSet(np.unique(encode(strs, k)), k)
