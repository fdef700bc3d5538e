"""The k-mer counting front end of the reference's ``scripts/better_consensus_unit_reconstruction.py`` on the GPU
(SURVEY.md §8(f) rank 2): ``get_kmer_counts_reads`` (:127-135) and ``get_most_frequent_kmers`` (:156-167), same
names and return shapes.  The de Bruijn graph purification and the edlib re-phasing that follow in the reference
(:170-190) are tiny host-side graph work and stay out of scope.
"""
import numpy as np

from . import kmers as km
from . import session


def get_kmer_counts_reads(ncrf_report, k=19):
    """kmer -> number of occurrences over all (de-gapped) read rows."""
    e = session.ensure_loaded(ncrf_report.packed, 1)
    e.count_occurrences(k)
    keys, lo, hi = e.table()
    return km.KmerFreqs(keys, lo.astype(np.int64) | (hi.astype(np.int64) << 32), k)


def get_most_frequent_kmers(reads_ncrf_report, k, unit_seq):
    """(counts mapping, set of the 3 * |unit k-mers| most frequent k-mers by (count, k-mer))."""
    unit_double_seq = unit_seq + unit_seq
    n_unit_kmers = len({unit_double_seq[i:i + k] for i in range(len(unit_seq))})
    kmer_counts_reads = get_kmer_counts_reads(reads_ncrf_report, k=k)
    keys, _ = session.engine().top_kmers(int(n_unit_kmers * 3))
    return kmer_counts_reads, set(km.decode(keys, k))
