"""Drop-in counterpart of the reference's ``scripts/cloud_contig.py``.

The pipeline's live use of this module — ``CloudContig.add_read`` and ``update_mapping_scores``
inside ``ReadPlacer.add_reads`` — runs on the GPU (cf_place_reads, centroflye_amd/read_placer.py).
This file keeps the reference's small-object API for interactive use and API parity (SURVEY.md §8
row A10: these helpers are not reachable from either CLI): plain-Python data structures with the
same attribute names and return values.  Reference: cloud_contig.py:8-41, :43-84, :87-156.
"""
from collections import Counter, defaultdict


class CloudContig:
    def __init__(self, min_cloud_kmer_freq):
        self.min_cloud_kmer_freq = max(1, min_cloud_kmer_freq)
        self.max_pos = 0
        self.clouds = defaultdict(Counter)       # position -> multiset of k-mers
        self.freq_clouds = defaultdict(set)      # position -> k-mers seen >= min_cloud_kmer_freq times there
        self.freq_kmers = set()
        self.kmer_positions = defaultdict(set)   # k-mer -> every position it was added at
        self.read_positions = {}
        self.coverage = defaultdict(int)

    def update_max_pos(self):
        self.max_pos = max(self.clouds) if self.clouds else 0

    def add_read(self, read_kmer_clouds, position):
        """Returns the (k-mer, position) pairs whose multiplicity just reached the threshold."""
        self.read_positions[read_kmer_clouds.r_id] = position
        fresh = []
        for pos, cloud in enumerate(read_kmer_clouds.kmers, start=position):
            self.coverage[pos] += 1
            counts = self.clouds[pos]
            for kmer in cloud:
                self.kmer_positions[kmer].add(pos)
                counts[kmer] += 1
                if counts[kmer] == self.min_cloud_kmer_freq:
                    self.freq_clouds[pos].add(kmer)
                    self.freq_kmers.add(kmer)
                    fresh.append((kmer, pos))
        self.update_max_pos()
        return fresh

    def calc_rough_inters_score(self, read_kmer_cloud):
        return len(set(read_kmer_cloud.all_kmers) & self.freq_kmers)

    def calc_inters_score(self, read_kmer_cloud, min_position=0, max_position=None, min_unit=2, min_inters=10, verbose=False):
        """Direct scoring: for every start position (units hit, shared k-mers); rightmost best wins."""
        if max_position is None:
            max_position = self.max_pos
        units = [set(c) for c in read_kmer_cloud.kmers]
        best_score, best_pos = (0, 0), None
        for pos in range(min_position, max_position + 1):
            hit = total = 0
            for i in range(min(self.max_pos - pos + 1, len(units))):
                shared = len(self.freq_clouds[pos + i] & units[i])
                hit += shared >= 1
                total += shared
            score = (hit, total)
            if hit >= min_unit and total >= min_inters and score >= best_score:
                best_score, best_pos = score, pos
        return best_score, best_pos

    def get_spread_kmers(self, max_npos=5):
        return {kmer for kmer in self.freq_kmers if len(self.kmer_positions[kmer]) > max_npos}


def update_mapping_scores(cloud_contig, kmers2pos, freq_kmers, scores=None):
    """scores[r_id][offset][unit] += 1 for every posting (r_id, unit) of every reported k-mer."""
    if scores is None:
        scores = defaultdict(lambda: defaultdict(Counter))
    for kmer, contig_pos in freq_kmers:
        for r_id, unit in kmers2pos.get(kmer, ()):
            if contig_pos >= unit:
                scores[r_id][contig_pos - unit][unit] += 1
    return scores


def _postings(reads_kmer_clouds):
    kmers2pos = defaultdict(list)
    for r_id in reads_kmer_clouds:
        for unit, cloud in enumerate(reads_kmer_clouds[r_id].kmers):
            for kmer in cloud:
                kmers2pos[kmer].append((r_id, unit))
    return kmers2pos


def map_reads(cloud_contig, reads_kmer_clouds, threshold=(5, 10), verbose=False):
    scores, pos = {}, {}
    for r_id in reads_kmer_clouds:
        cloud = reads_kmer_clouds[r_id]
        best_score, best_pos = cloud_contig.calc_inters_score(cloud, max_position=cloud_contig.max_pos - len(cloud.kmers) + 1)
        if best_pos == 0 or best_score > threshold:
            scores[r_id], pos[r_id] = best_score, best_pos
    return pos, scores


def map_reads_fast(cloud_contig, reads_kmer_clouds, threshold=(5, 10), verbose=False, debug=False):
    seed = [(kmer, p) for kmer in cloud_contig.freq_kmers for p in cloud_contig.kmer_positions[kmer]]
    scores = update_mapping_scores(cloud_contig, _postings(reads_kmer_clouds), seed)
    positions = {}
    for r_id in reads_kmer_clouds:
        n_units = len(reads_kmer_clouds[r_id].kmers)
        best = None
        for pos, per_unit in scores[r_id].items():
            if pos + n_units > len(cloud_contig.clouds):
                continue
            score = (len(per_unit), sum(per_unit.values()))
            if score[0] < threshold[0] or score[1] < threshold[1]:
                continue
            if best is None or (score, pos) > best:
                best = (score, pos)
        if best is not None:
            positions[r_id] = best[1]
    return positions, scores
