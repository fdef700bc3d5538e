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


def edge_lines_digest(lines):
    return sha(sorted(lines))
