"""Drop-in counterpart of the reference's ``scripts/read_kmer_cloud.py``.

``get_reads_kmer_clouds`` (reference :34-40) and ``filter_reads_kmer_clouds`` (:43-54) run on
the GPU (cf_build_clouds / cf_filter_clouds) and return an array-backed mapping
``r_id -> ReadKMerCloud`` whose per-read objects are materialised on access.
"""
import itertools
import math
from collections.abc import Mapping

import numpy as np

from . import kmers as km
from . import session


class ReadKMerCloud:
    """Per read: list (one per HOR unit) of sets of k-mer strings (reference :9-15)."""

    def __init__(self, kmers, r_id):
        self.r_id = r_id
        self.kmers = kmers
        self.all_kmers = list(itertools.chain.from_iterable(kmers))

    @classmethod
    def fromNCRF_record(cls, ncrf_record, n, k, genomic_kmers):
        clouds = get_reads_kmer_clouds(ncrf_record._report, n=n, k=k, genomic_kmers=genomic_kmers)
        return clouds[ncrf_record.r_id]


class KMerClouds(Mapping):
    """r_id -> ReadKMerCloud over CSR arrays (unit_ptr per read, cloud_ptr per unit, entries =
    indices into the sorted k-mer set)."""

    def __init__(self, report, n, kset, unit_ptr, cloud_ptr, entries):
        self.report, self.n, self.kset = report, int(n), kset
        self.unit_ptr, self.cloud_ptr, self.entries = unit_ptr, cloud_ptr, entries
        self._row = {r_id: i for i, r_id in enumerate(report.packed.ids)}
        self._strings = None
        self.token = object()

    def __len__(self):
        return len(self._row)

    def __iter__(self):
        return iter(self.report.packed.ids)

    def __getitem__(self, r_id):
        r = self._row[r_id]
        if self._strings is None:
            self._strings = np.array(self.kset.strings(), dtype=object)
        cp, ent = self.cloud_ptr, self.entries
        units = [set(self._strings[ent[cp[u]:cp[u + 1]]]) for u in range(self.unit_ptr[r], self.unit_ptr[r + 1])]
        return ReadKMerCloud(units, r_id)

    def on_device(self):
        """Engine holding exactly these clouds (re-installs them if something else ran since)."""
        e = session.ensure_loaded(self.report.packed, self.n)
        if session.clouds_token() is not self.token:
            e.set_kmers(self.kset.codes, self.kset.k)
            e.set_clouds(self.cloud_ptr, self.entries)
            session.set_clouds_token(self.token)
        return e


_UPPER = np.arange(256, dtype=np.uint8)
_UPPER[ord("a"):ord("z") + 1] -= 32
_NOT_ACGT = np.ones(256, bool)
_NOT_ACGT[list(b"ACGT")] = False


def exotic_hits(packed, n, kset):
    """(unit, rank) pairs, sorted and unique: the units one of whose upper-cased windows (reference :25-29) is a k-mer of
    ``kset.extra`` — the k-mers the 2-bit lookup of cf_build_clouds cannot find.  Only windows over a symbol that is not A, C, G, T
    after upper-casing can be one; such symbols are rare (N calls), so this stays on the host."""
    k = kset.k
    _, us, ue, _ = packed.units(n)
    up = _UPPER[np.asarray(packed.bases)]
    bad = np.flatnonzero(_NOT_ACGT[up])
    hits = set()
    if bad.size and len(us):
        u = np.searchsorted(us, bad, side="right") - 1
        ok = (u >= 0) & (bad < ue[np.maximum(u, 0)])
        for b, uu in zip(bad[ok].tolist(), u[ok].tolist()):
            lo, hi = max(int(us[uu]), b - k + 1), min(b, int(ue[uu]) - k)
            if hi < lo:
                continue
            text = up[lo:hi + k].tobytes().decode("latin-1")
            for w in range(hi - lo + 1):
                r = kset._extra_rank.get(text[w:w + k])
                if r is not None:
                    hits.add((uu, r))
    hits = sorted(hits)
    return np.array([h[0] for h in hits], np.int64), np.array([h[1] for h in hits], np.int64)


def add_exotic_entries(packed, n, kset, cloud_ptr, entries):
    """The cloud CSR with the entries of exotic_hits added (their ranks lie behind every 2-bit k-mer's: the end of the row)."""
    hu, hr = exotic_hits(packed, n, kset)
    if not hu.size:
        return cloud_ptr, entries
    cloud_ptr = np.asarray(cloud_ptr, np.int64)
    entries = np.insert(np.asarray(entries), cloud_ptr[hu + 1], hr.astype(np.asarray(entries).dtype))
    grow = np.concatenate([[0], np.cumsum(np.bincount(hu, minlength=cloud_ptr.size - 1))]).astype(np.int64)
    return cloud_ptr + grow, entries


def get_reads_kmer_clouds(ncrf_report, n, k, genomic_kmers=None):
    kset = km.as_kmer_set(genomic_kmers, k)
    e = session.ensure_loaded(ncrf_report.packed, n)
    e.set_kmers(kset.codes, k)
    e.build_clouds()
    cloud_ptr, entries = e.clouds()
    if kset.extra:      # k-mers with an N (...): the device finds none of them; their few cloud entries come from the host
        cloud_ptr, entries = add_exotic_entries(ncrf_report.packed, n, kset, cloud_ptr, entries)
        e.set_clouds(cloud_ptr, entries)
    unit_ptr = ncrf_report.packed.units(n)[0]
    out = KMerClouds(ncrf_report, n, kset, np.array(unit_ptr), cloud_ptr, entries)
    session.set_clouds_token(out.token)
    return out


def filter_reads_kmer_clouds(kmer_clouds, min_mult=2, max_mult=math.inf):
    e = kmer_clouds.on_device()
    e.filter_clouds(int(min_mult), 0 if max_mult == math.inf else int(max_mult))
    kmer_clouds.cloud_ptr, kmer_clouds.entries = e.clouds()
    kmer_clouds.token = object()
    session.set_clouds_token(kmer_clouds.token)
    return kmer_clouds


def get_all_kmers(kmer_clouds):
    """Sorted list of all cloud k-mers (the reference's version, :57-63, raises on its own typo)."""
    out = sorted(itertools.chain.from_iterable(kmer_clouds[r].all_kmers for r in kmer_clouds))
    return out
