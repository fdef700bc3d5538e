"""Host-side small-object counterpart of the reference's ``scripts/cloud_contig.py``.

Not on the pipeline's path: the live use of the cloud contig (adding reads and updating mapping scores inside
``ReadPlacer.add_reads``) runs on the GPU (``cf_place_reads``, see ``centroflye_amd/read_placer.py``).  This module
exists for interactive use and API parity: the same public names, attributes and return
values as the reference (cloud_contig.py:8-41 CloudContig, :43-84 scoring helpers, :87-95 update_mapping_scores,
:98-156 map_reads / map_reads_fast), written on plain dict / Counter containers.
"""
from collections import Counter, defaultdict
from itertools import chain


def _score_of(per_unit_hits):
    """(number of read units with at least one hit, total hits)."""
    return (sum(1 for h in per_unit_hits if h > 0), sum(per_unit_hits))


class CloudContig:
    """Per contig position a multiset of k-mers; a k-mer becomes *frequent* at a position when its multiplicity
    there reaches ``min_cloud_kmer_freq`` (at least 1)."""

    def __init__(self, min_cloud_kmer_freq):
        self.min_cloud_kmer_freq = min_cloud_kmer_freq if min_cloud_kmer_freq > 1 else 1
        self.clouds = defaultdict(Counter)
        self.freq_clouds = defaultdict(set)
        self.kmer_positions = defaultdict(set)
        self.coverage = defaultdict(int)
        self.freq_kmers = set()
        self.read_positions = {}
        self.max_pos = 0

    def update_max_pos(self):
        self.max_pos = max(self.clouds, default=0)

    def add_read(self, read_kmer_clouds, position):
        """Lay the read's unit clouds onto positions position, position+1, ...; returns the (k-mer, position) pairs
        that turned frequent through this read (each at most once)."""
        turned = []
        threshold = self.min_cloud_kmer_freq
        self.read_positions[read_kmer_clouds.r_id] = position
        for offset, cloud in enumerate(read_kmer_clouds.kmers):
            where = position + offset
            self.coverage[where] += 1
            column = self.clouds[where]
            column.update(cloud)
            for kmer in cloud:
                self.kmer_positions[kmer].add(where)
                if column[kmer] == threshold:     # a set contributes each k-mer once, so equality fires exactly once
                    turned.append((kmer, where))
        for kmer, where in turned:
            self.freq_clouds[where].add(kmer)
            self.freq_kmers.add(kmer)
        self.update_max_pos()
        return turned

    def calc_rough_inters_score(self, read_kmer_cloud):
        return sum(1 for kmer in set(read_kmer_cloud.all_kmers) if kmer in self.freq_kmers)

    def calc_inters_score(self, read_kmer_cloud, min_position=0, max_position=None, min_unit=2, min_inters=10, verbose=False):
        """Brute-force placement score of one read: for each start position the pair (units hit, shared frequent
        k-mers); among positions passing both thresholds the best pair wins, the rightmost on ties."""
        last = self.max_pos if max_position is None else max_position
        unit_sets = [frozenset(c) for c in read_kmer_cloud.kmers]
        winner, winner_pos = (0, 0), None
        for start in range(min_position, last + 1):
            span = min(len(unit_sets), self.max_pos - start + 1)
            candidate = _score_of([len(unit_sets[i] & self.freq_clouds[start + i]) for i in range(span)])
            if candidate[0] >= min_unit and candidate[1] >= min_inters and candidate >= winner:
                winner, winner_pos = candidate, start
        return winner, winner_pos

    def get_spread_kmers(self, max_npos=5):
        return set(k for k in self.freq_kmers if len(self.kmer_positions[k]) > max_npos)


def update_mapping_scores(cloud_contig, kmers2pos, freq_kmers, scores=None):
    """Posting-list join: a frequent k-mer at contig position q supports, for each (read, unit) holding it, the
    hypothesis that the read starts at q - unit.  ``scores[read][start][unit]`` counts the support."""
    scores = defaultdict(lambda: defaultdict(Counter)) if scores is None else scores
    for kmer, q in freq_kmers:
        for r_id, unit in kmers2pos.get(kmer, ()):
            start = q - unit
            if start >= 0:
                scores[r_id][start][unit] += 1
    return scores


def _inverted_index(reads_kmer_clouds):
    index = defaultdict(list)
    for r_id in reads_kmer_clouds:
        for unit, cloud in enumerate(reads_kmer_clouds[r_id].kmers):
            for kmer in cloud:
                index[kmer].append((r_id, unit))
    return index


def map_reads(cloud_contig, reads_kmer_clouds, threshold=(5, 10), verbose=False):
    """Place every read with calc_inters_score; keeps reads that land at 0 or beat the threshold pair."""
    pos, scores = {}, {}
    for r_id in reads_kmer_clouds:
        read = reads_kmer_clouds[r_id]
        score, where = cloud_contig.calc_inters_score(read, max_position=cloud_contig.max_pos - len(read.kmers) + 1)
        if where == 0 or score > threshold:
            pos[r_id], scores[r_id] = where, score
    return pos, scores


def map_reads_fast(cloud_contig, reads_kmer_clouds, threshold=(5, 10), verbose=False, debug=False):
    """Same placement through the posting-list scores (the contig's current frequent k-mers at all their positions)."""
    seed = list(chain.from_iterable(((kmer, q) for q in cloud_contig.kmer_positions[kmer]) for kmer in cloud_contig.freq_kmers))
    scores = update_mapping_scores(cloud_contig, _inverted_index(reads_kmer_clouds), seed)
    n_positions = len(cloud_contig.clouds)
    positions = {}
    for r_id in reads_kmer_clouds:
        length = len(reads_kmer_clouds[r_id].kmers)
        ranked = [((len(units), sum(units.values())), start) for start, units in scores[r_id].items() if start + length <= n_positions]
        ranked = [x for x in ranked if x[0][0] >= threshold[0] and x[0][1] >= threshold[1]]
        if ranked:
            positions[r_id] = max(ranked)[1]
    return positions, scores
