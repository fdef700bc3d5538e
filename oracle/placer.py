"""A4, A8, A9 — cloud multiplicity filter and greedy cloud-contig read placement.

Restates, on CSR clouds of integer k-mer indices:
  A4 filter_reads_kmer_clouds        read_kmer_cloud.py:43-54
  A8 CloudContig.add_read            cloud_contig.py:26-41   (events when a count EQUALS the threshold)
     update_mapping_scores           cloud_contig.py:87-95
  A9 ReadPlacer.add_prefix_reads     read_placer.py:35-40
     ReadPlacer.add_reads            read_placer.py:42-94    (seed over-inclusion :54-57, tie-breaks :68-78)
     ReadPlacer.run                  read_placer.py:96-128
Scores are kept as (s0, s1) per (read, offset) plus a seen-set of (read, offset, unit) instead
of the reference's dict-of-dict-of-Counter; s0 = #units hit, s1 = total hits.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np


def filter_clouds(cloud_ptr, entries, n_kmers, min_mult=2, max_mult=None):
    """Keep in every cloud only k-mers present in >= min_mult (<= max_mult) clouds overall."""
    cnt = np.bincount(entries, minlength=max(n_kmers, 1))
    keep = cnt[entries] >= min_mult
    if max_mult is not None:
        keep &= cnt[entries] <= max_mult
    csum = np.concatenate([[0], np.cumsum(keep)]).astype(np.int64)
    per = csum[cloud_ptr[1:]] - csum[cloud_ptr[:-1]]
    new_ptr = np.concatenate([[0], np.cumsum(per)]).astype(np.int64)
    return new_ptr, entries[keep]


class _Contig:
    def __init__(self, min_cloud_kmer_freq):
        self.thr = max(1, int(min_cloud_kmer_freq))
        self.count = {}       # (pos, kmer) -> multiplicity
        self.freq = set()     # k-mers frequent at some position
        self.positions = {}   # kmer -> set of positions it was ever added at

    def add(self, clouds, position):
        events = []
        for i, cloud in enumerate(clouds):
            q = position + i
            for x in cloud:
                self.positions.setdefault(x, set()).add(q)
                c = self.count.get((q, x), 0) + 1
                self.count[(q, x)] = c
                if c == self.thr:
                    self.freq.add(x)
                    events.append((x, q))
        return events


def place_reads(ids, classes, unit_ptr, cloud_ptr, entries, min_cloud_kmer_freq=2,
                min_unit=2, min_inters=10, min_prop=3):
    """classes: 0 prefix, 1 internal, 2 suffix per read (record order).
    Returns the list of lines of read_positions.csv in the order the reference writes them,
    except that each stage's trailing 'r_id None' block is sorted by r_id (the reference's
    order there is set-iteration order)."""
    ent = entries.tolist()
    cp = cloud_ptr.tolist()
    up = unit_ptr.tolist()

    def clouds_of(r):
        return [ent[cp[u]:cp[u + 1]] for u in range(up[r], up[r + 1])]

    contig = _Contig(min_cloud_kmer_freq)
    lines = []
    for r in np.flatnonzero(classes == 0):
        contig.add(clouds_of(r), 0)
        lines.append(f"{ids[r]} 0")
    for cls in (1, 2):
        stage = [int(r) for r in np.flatnonzero(classes == cls)]
        postings = {}
        for r in stage:
            for i, cloud in enumerate(clouds_of(r)):
                for x in cloud:
                    postings.setdefault(x, []).append((r, i))
        unused = set(stage)
        s0, s1, seen = {}, {}, set()
        events = [(x, q) for x in contig.freq for q in contig.positions[x]]
        while unused:
            for x, q in events:
                for r, i in postings.get(x, ()):
                    if q >= i:
                        key = (r, q - i)
                        s1[key] = s1.get(key, 0) + 1
                        if (r, q - i, i) not in seen:
                            seen.add((r, q - i, i))
                            s0[key] = s0.get(key, 0) + 1
            best = None
            for (r, off), v1 in s1.items():
                if r not in unused:
                    continue
                v0 = s0[(r, off)]
                if v0 >= min_unit and v0 * min_prop <= v1 and v1 >= min_inters:
                    cand = (v0, v1, off)
                    if best is None or cand > best[0] or (cand == best[0] and ids[r] < ids[best[1]]):
                        best = (cand, r)
            if best is None:
                lines.extend(sorted(f"{ids[r]} None" for r in unused))
                break
            (v0, v1, off), r = best
            lines.append(f"{ids[r]} {off} {v0} {v1}")
            events = contig.add(clouds_of(r), off)
            unused.discard(r)
    return lines


def stage3(records, all_alns, read_lens, genomic_kmers_sorted, n_motif=1, k_cloud=19,
           min_cloud_kmer_freq=2, min_kmer_mult=2, min_unit=2, min_inters=10, prefix_threshold=50000):
    """End-to-end restatement of ReadPlacer(params).run() on oracle records."""
    from . import ncrf, recruit
    recs = list(records.values())
    ids = [r.r_id for r in recs]
    pre, mid, suf = ncrf.classify(records, all_alns, read_lens, prefix_threshold)
    cls = np.ones(len(ids), np.int64)
    pre_s, suf_s = set(pre), set(suf)
    for i, r_id in enumerate(ids):
        if r_id in pre_s:
            cls[i] = 0
        elif r_id in suf_s:
            cls[i] = 2
    units = [[s.encode() for s in ncrf.unit_reads(r, n_motif)] for r in recs]
    unit_ptr, cloud_ptr, entries = recruit.build_clouds(units, k_cloud, genomic_kmers_sorted)
    f_ptr, f_ent = filter_clouds(cloud_ptr, entries, genomic_kmers_sorted.size, min_kmer_mult)
    lines = place_reads(ids, cls, unit_ptr, f_ptr, f_ent, min_cloud_kmer_freq, min_unit, min_inters)
    return dict(classes=cls, unit_ptr=unit_ptr, cloud_ptr=cloud_ptr, entries=entries,
                f_cloud_ptr=f_ptr, f_entries=f_ent, lines=lines)
