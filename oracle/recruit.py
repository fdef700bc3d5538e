"""A1-A7 — rare k-mer recruitment, k-mer clouds, distance histogram, edge filter (numpy).

Array-based restatement of the reference's stage 2:
  A1 presence counts with multi-occurrence cut  distance_based_kmer_recruitment.py:39-63
  A2 rare window                                distance_based_kmer_recruitment.py:66-82
  A3 per-unit clouds                            read_kmer_cloud.py:17-40
  A5 (a, b, d) co-occurrence histogram          distance_based_kmer_recruitment.py:85-128
  A6 min-coverage + 0.8 dominance filter        distance_based_kmer_recruitment.py:131-149
  A7 output files                               distance_based_kmer_recruitment.py:152-171
k-mers are 2-bit codes (A<C<G<T, first base most significant), so integer order equals the
string order the reference sorts by.  Inputs must be upper-case ACGT.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np

_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def encode_windows(seq, k):
    """uint64 codes of all k-long windows of seq (bytes / uint8 array of ACGT)."""
    b = np.frombuffer(seq, dtype=np.uint8) if isinstance(seq, (bytes, bytearray)) else np.asarray(seq, dtype=np.uint8)
    n = b.size - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    two = _CODE[b]
    if (two == 255).any():
        raise ValueError("non-ACGT base: outside the oracle's alphabet")
    two = two.astype(np.uint64)
    out = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        out |= two[j:j + n] << np.uint64(2 * (k - 1 - j))
    return out


def decode_kmer(code, k):
    code = int(code)
    return "".join("ACGT"[(code >> (2 * (k - 1 - j))) & 3] for j in range(k))


def encode_kmer(s):
    v = 0
    for ch in s:
        v = (v << 2) | "ACGT".index(ch)
    return v


# ------------------------------------------------------------------ A1
def kmer_presence(read_seqs, k, max_nonuniq):
    """read_seqs: iterable of de-gapped read rows (bytes).  Returns (keys, pres) for the k-mers
    that survive the multi-occurrence cut: pres = #reads containing the k-mer, dropped when it
    occurs >= 2x inside more than max_nonuniq reads.  Also returns the counters dict."""
    keys_l, multi_l = [], []
    n_w = n_rk = 0
    for s in read_seqs:
        codes = encode_windows(s, k)
        n_w += codes.size
        u, c = np.unique(codes, return_counts=True)
        n_rk += u.size
        keys_l.append(u)
        multi_l.append(c > 1)
    if not keys_l:
        z = np.zeros(0, dtype=np.uint64)
        return z, np.zeros(0, dtype=np.int64), dict(n_w=0, n_rk=0, n_distinct=0)
    allk = np.concatenate(keys_l)
    allm = np.concatenate(multi_l)
    order = np.argsort(allk, kind="stable")
    allk, allm = allk[order], allm[order]
    keys, start, pres = np.unique(allk, return_index=True, return_counts=True)
    multi = np.add.reduceat(allm.astype(np.int64), start) if keys.size else np.zeros(0, np.int64)
    keep = multi <= max_nonuniq
    return keys[keep], pres[keep].astype(np.int64), dict(n_w=n_w, n_rk=n_rk, n_distinct=int(keys.size))


# ------------------------------------------------------------------ A2
def rare_bounds(bottom, top, coverage, kmer_survival_rate):
    """Integer [lo, hi] equivalent of `left <= freq <= right` with the reference's own double
    expressions (distance_based_kmer_recruitment.py:74-78)."""
    left = bottom * coverage * kmer_survival_rate
    right = top * coverage * kmer_survival_rate
    lo = int(np.ceil(left))
    while lo - 1 >= left:
        lo -= 1
    while lo < left:
        lo += 1
    hi = int(np.floor(right))
    while hi + 1 <= right:
        hi += 1
    while hi > right:
        hi -= 1
    return max(lo, 0), hi


def select_rare(keys, pres, lo, hi):
    """Sorted rare k-mer codes."""
    m = (pres >= lo) & (pres <= hi)
    return np.sort(keys[m])


# ------------------------------------------------------------------ A3
def build_clouds(unit_seqs_per_read, k, kmers_sorted):
    """unit_seqs_per_read: list (per read) of lists (per unit) of de-gapped upper-case unit
    rows.  Returns CSR (unit_ptr[R+1], cloud_ptr[U+1], entries int32 sorted-unique indices into
    kmers_sorted)."""
    unit_ptr = [0]
    cloud_ptr = [0]
    ent = []
    for units in unit_seqs_per_read:
        for s in units:
            codes = np.unique(encode_windows(s, k))
            idx = np.searchsorted(kmers_sorted, codes)
            idx[idx >= kmers_sorted.size] = 0
            hit = kmers_sorted[idx] == codes if kmers_sorted.size else np.zeros(codes.size, bool)
            ent.append(idx[hit].astype(np.int32))
            cloud_ptr.append(cloud_ptr[-1] + int(hit.sum()))
        unit_ptr.append(len(cloud_ptr) - 1)
    entries = np.concatenate(ent) if ent else np.zeros(0, np.int32)
    return np.array(unit_ptr, np.int64), np.array(cloud_ptr, np.int64), entries


# ------------------------------------------------------------------ A5
def dist_histogram(unit_ptr, cloud_ptr, entries, n_kmers, min_n, max_n, min_d, max_d):
    """Sparse histogram over (a, b, d): returns (a, b, d, cnt) arrays sorted by (a, b, d) and the
    number of pair emissions E.  Reads [min_n, max_n) in record order; for d in [min_d, max_d]
    unit pairs (i, i+d) with i < U - d (Python slice kmer_clouds[:-d]; empty for d = 0)."""
    R = unit_ptr.size - 1
    lo, hi = max(0, min_n), min(R, max_n)
    D = max_d + 1
    chunks = []
    E = 0
    acc_keys = np.zeros(0, np.uint64)
    acc_cnt = np.zeros(0, np.int64)
    pend, pend_n = [], 0

    def flush():
        nonlocal acc_keys, acc_cnt, pend, pend_n
        if not pend:
            return
        allk = np.concatenate([acc_keys] + pend)
        w = np.concatenate([acc_cnt] + [np.ones(p.size, np.int64) for p in pend])
        order = np.argsort(allk, kind="stable")
        allk, w = allk[order], w[order]
        start = np.flatnonzero(np.concatenate(([True], allk[1:] != allk[:-1]))) if allk.size else np.zeros(0, np.int64)      # (allk is sorted: run starts)
        acc_keys = allk[start]
        acc_cnt = np.add.reduceat(w, start) if acc_keys.size else np.zeros(0, np.int64)
        pend, pend_n = [], 0

    nk = np.uint64(max(n_kmers, 1))
    for r in range(lo, hi):
        u0, u1 = int(unit_ptr[r]), int(unit_ptr[r + 1])
        U = u1 - u0
        for d in range(max(min_d, 1), max_d + 1):
            if d >= U:
                break
            for i in range(U - d):
                A = entries[cloud_ptr[u0 + i]:cloud_ptr[u0 + i + 1]].astype(np.uint64)
                B = entries[cloud_ptr[u0 + i + d]:cloud_ptr[u0 + i + d + 1]].astype(np.uint64)
                if A.size == 0 or B.size == 0:
                    continue
                aa = np.repeat(A, B.size)
                bb = np.tile(B, A.size)
                ne = aa != bb
                key = (aa[ne] * nk + bb[ne]) * np.uint64(D) + np.uint64(d)
                E += key.size
                pend.append(key)
                pend_n += key.size
                if pend_n > 20_000_000:
                    flush()
    flush()
    d_arr = (acc_keys % np.uint64(D)).astype(np.int64)
    ab = acc_keys // np.uint64(D)
    return (ab // nk).astype(np.int64), (ab % nk).astype(np.int64), d_arr, acc_cnt, E


# ------------------------------------------------------------------ A6
def filter_edges(a, b, d, cnt, min_coverage, rel_threshold=0.8):
    """Edges (d, a, b, cnt) with cnt >= min_coverage and cnt / sum_d cnt(a,b,.) >= rel_threshold
    (true division in doubles, as Python's int / int).  Input sorted by (a, b, d).
    Returns (edges int64[n,4] sorted by (d, a, b), unique k-mer indices sorted)."""
    if a.size == 0:
        return np.zeros((0, 4), np.int64), np.zeros(0, np.int64)
    new = np.ones(a.size, bool)
    new[1:] = (a[1:] != a[:-1]) | (b[1:] != b[:-1])
    grp = np.cumsum(new) - 1
    tot = np.bincount(grp, weights=cnt.astype(np.float64)).astype(np.int64)[grp]
    sel = (cnt >= min_coverage) & (cnt.astype(np.float64) / tot.astype(np.float64) >= rel_threshold)
    e = np.stack([d[sel], a[sel], b[sel], cnt[sel]], axis=1).astype(np.int64)
    order = np.lexsort((e[:, 2], e[:, 1], e[:, 0]))
    e = e[order]
    uniq = np.unique(np.concatenate([e[:, 1], e[:, 2]])) if e.size else np.zeros(0, np.int64)
    return e, uniq


# ------------------------------------------------------------------ A7
def kmers_file_text(kmers_sorted, unique_idx, k):
    """Byte-exact content of unique_kmers_min_edge_cov_N.txt (sorted k-mer strings)."""
    return "".join(decode_kmer(kmers_sorted[i], k) + "\n" for i in np.sort(unique_idx))


def edges_file_lines(kmers_sorted, edges, k):
    """Lines of unique_edges_min_edge_cov_N.txt ('d kmer_a kmer_b cnt'); the reference's line
    order is hash-seed dependent, so parity is on the sorted list."""
    return sorted(f"{int(e[0])} {decode_kmer(kmers_sorted[e[1]], k)} {decode_kmer(kmers_sorted[e[2]], k)} {int(e[3])}"
                  for e in edges)


def stage2(records, k=19, bottom=0.9, top=3.0, coverage=32, kmer_survival_rate=0.34, max_nonuniq=3,
           min_n=0, max_n=2 ** 62, min_d=1, max_d=150, min_coverage=4, n_motif=1):
    """End-to-end restatement of distance_based_kmer_recruitment.main (:174-208) on oracle
    records (oracle.ncrf.parse_report).  Returns a dict of every intermediate."""
    from . import ncrf
    recs = list(records.values())
    seqs = [r.r_al.replace("-", "").encode() for r in recs]
    keys, pres, counters = kmer_presence(seqs, k, max_nonuniq)
    lo, hi = rare_bounds(bottom, top, coverage, kmer_survival_rate)
    rare = select_rare(keys, pres, lo, hi)
    units = [[s.encode() for s in ncrf.unit_reads(r, n_motif)] for r in recs]
    unit_ptr, cloud_ptr, entries = build_clouds(units, k, rare)
    a, b, d, cnt, E = dist_histogram(unit_ptr, cloud_ptr, entries, rare.size, min_n, max_n, min_d, max_d)
    edges, uniq = filter_edges(a, b, d, cnt, min_coverage)
    counters.update(n_b=sum(len(s) for s in seqs), n_rare=int(rare.size), n_ce=int(entries.size), E=int(E),
                    n_edges=int(edges.shape[0]), n_unique=int(uniq.size), lo=lo, hi=hi)
    return dict(keys=keys, pres=pres, rare=rare, unit_ptr=unit_ptr, cloud_ptr=cloud_ptr, entries=entries,
                hist=(a, b, d, cnt), edges=edges, unique=uniq, counters=counters)
