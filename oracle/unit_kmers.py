"""SURVEY.md §8(f) rank 2 — k-mer occurrence counts and the most frequent k-mers that seed the reference's unit
reconstruction (numpy restatement).

  scripts/better_consensus_unit_reconstruction.py:127-135  get_kmer_counts_reads: every window of every de-gapped
      read row counts once (no per-read de-duplication, no upper-casing)
  :156-167  get_most_frequent_kmers: n = 3 * |{k-mers of the circular unit}|, the n k-mers with the largest
      (count, k-mer string)
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

from .recruit import encode_windows


def kmer_occurrences(read_seqs, k):
    """-> (sorted distinct codes, occurrence counts)."""
    parts = [encode_windows(s, k) for s in read_seqs]
    allc = np.concatenate(parts) if parts else np.zeros(0, np.uint64)
    return np.unique(allc, return_counts=True)


def n_circular_unit_kmers(unit_seq, k):
    dbl = unit_seq + unit_seq
    return len({dbl[i:i + k] for i in range(len(unit_seq))})


def most_frequent(keys, counts, n):
    """indices into keys of the n largest by (count, code), in descending order."""
    order = np.lexsort((keys, counts))[::-1]
    return order[:n]
