"""k-mer strings <-> 2-bit codes (A=0 C=1 G=2 T=3, first base most significant) and the
array-backed containers the drop-in modules hand out in place of the reference's dict / set of
Python strings (they keep the same mapping / set protocol for small inputs).

k-mers the 2-bit code does not have — an N or another upper-case symbol in the window, no lower-case letter — are carried beside
the codes as strings (KmerSet.extra): the reference selects, finds and writes them like any other k-mer
(distance_based_kmer_recruitment.py:47-53, read_kmer_cloud.py:25).  On the device each of them is a pseudo-code EXOTIC_BASE + j behind
every real code: it takes a rank, a posting list, edges — and never matches a window by itself; the host adds their cloud entries
(read_kmer_cloud.py here)."""
import itertools
from collections.abc import Mapping, Set

import numpy as np

_LUT = np.full(256, 255, np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _LUT[_c] = _i
_ALPHA = np.frombuffer(b"ACGT", np.uint8)
EXOTIC_BASE = 1 << 62      # codes of k <= 31 stay below 2^62; bit 63 is the device table's occupancy bit


def is_exotic(s, k):
    """A k-long string that is no ACGT word but could equal a window of an upper-cased unit: no lower-case letter, no white space."""
    return isinstance(s, str) and len(s) == k and try_encode(s, k) is None and s == s.upper() and not any(c.isspace() for c in s)


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
    """Sorted unique k-mer codes behaving like the reference's ``set`` of k-mer strings.  ``extra``: the k-mers without a 2-bit
    code (sorted strings); ``codes`` ends with one pseudo-code per extra k-mer, so ranks run over both."""

    def __init__(self, codes, k, extra=(), inert=()):
        codes = np.ascontiguousarray(codes, np.uint64)
        self.k = int(k)
        self.extra = tuple(sorted(set(extra)))
        # members only: k-mers with a lower-case letter (rare windows of soft-masked stretches — the reference's get_rare_kmers returns them,
        # no cloud can hold them); they have no code, their indices lie behind every rank
        self.inert = tuple(sorted(set(inert)))
        if codes.size and int(codes[-1]) >= EXOTIC_BASE:      # (already carries its pseudo-codes: the device's list read back)
            codes = codes[codes < np.uint64(EXOTIC_BASE)]
        self.n_acgt = int(codes.size)
        if self.extra:
            codes = np.concatenate([codes, np.uint64(EXOTIC_BASE) + np.arange(len(self.extra), dtype=np.uint64)])
        self.codes = np.ascontiguousarray(codes, np.uint64)
        self._extra_rank = {s: self.n_acgt + j for j, s in enumerate(self.extra)}
        self._inert_rank = {s: int(self.codes.size) + j for j, s in enumerate(self.inert)}

    @classmethod
    def _from_iterable(cls, it):   # results of set algebra fall back to plain sets
        return set(it)

    def __len__(self):
        return int(self.codes.size) + len(self.inert)

    def __contains__(self, s):
        if s in self._extra_rank or s in self._inert_rank:
            return True
        c = try_encode(s, self.k)
        if c is None:
            return False
        i = int(np.searchsorted(self.codes[:self.n_acgt], np.uint64(c)))
        return i < self.n_acgt and int(self.codes[i]) == c

    def strings(self, ranks=None):
        """the k-mers of the given ranks (all, in rank order, by default) as strings"""
        if ranks is None:
            return decode(self.codes[:self.n_acgt], self.k) + list(self.extra) + list(self.inert)
        ranks = np.asarray(ranks, np.int64)
        out = np.empty(ranks.size, dtype=object)
        plain = ranks < self.n_acgt
        if plain.any():
            out[plain] = decode(self.codes[ranks[plain]], self.k)
        for i in np.flatnonzero(~plain):
            j = int(ranks[i]) - self.n_acgt
            out[i] = self.extra[j] if j < len(self.extra) else self.inert[j - len(self.extra)]
        return out.tolist()

    def __iter__(self):
        return iter(self.strings())

    def index(self, s):
        if s in self._extra_rank:
            return self._extra_rank[s]
        if s in self._inert_rank:
            return self._inert_rank[s]
        c = try_encode(s, self.k)
        i = int(np.searchsorted(self.codes[:self.n_acgt], np.uint64(c))) if c is not None else -1
        if c is None or i >= self.n_acgt or int(self.codes[i]) != c:
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
    """kmer string -> number of reads containing it (the reference's ``all_kmers`` dict).  ``extra``: the keys without a 2-bit code
    (windows with an N, soft-masked stretches: the reference counts the raw text) as a plain dict."""

    def __init__(self, codes, pres, k, extra=None):
        self.codes = np.ascontiguousarray(codes, np.uint64)
        self.pres = np.ascontiguousarray(pres, np.int64)
        self.k = int(k)
        self.extra = dict(extra or {})

    def __len__(self):
        return int(self.codes.size) + len(self.extra)

    def __iter__(self):
        return itertools.chain(decode(self.codes, self.k), self.extra)

    def __getitem__(self, s):
        if s in self.extra:
            return self.extra[s]
        c = try_encode(s, self.k)
        i = int(np.searchsorted(self.codes, np.uint64(c))) if c is not None else -1
        if c is None or i >= self.codes.size or int(self.codes[i]) != c:
            raise KeyError(s)
        return int(self.pres[i])

    def items(self):
        return itertools.chain(zip(decode(self.codes, self.k), (int(v) for v in self.pres)), self.extra.items())


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
    kmers = list(kmers)
    strs = [s for s in kmers if try_encode(s, k) is not None]
    # other strings match a window of an upper-cased unit only if they hold no lower-case letter (read_kmer_cloud.py:25)
    return KmerSet(np.unique(encode(strs, k)), k, [s for s in kmers if is_exotic(s, k)])


def exotic_lines(path, k):
    """The lines of a k-mer file (reference read_placer.py:23-25: every stripped line is a k-mer) that are k long, hold a symbol
    other than A, C, G, T and no lower-case letter; the 2-bit reader (cfh_read_kmers) skips them."""
    with open(path, "rb") as f:
        if not f.read().translate(None, b"ACGT\r\n"):      # (nothing but k-mers of the 2-bit alphabet: the usual file)
            return []
    with open(path) as f:
        return sorted({ln.strip() for ln in f if is_exotic(ln.strip(), k)})
