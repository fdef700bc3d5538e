"""A0 — NCRF report parsing, orientation, unit split, classification (numpy/pure Python).

Restates scripts/ncrf_parser.py of the reference:
  record selection + orientation  :61-118
  unit split                      :28-59   (regex 'base([-]*)' per motif base, motif*n)
  classify                        :120-145
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from collections import OrderedDict

_COMP = str.maketrans("ATGCatgc-", "TACGtacg-")  # utils/bio.py:27-29


def revcomp(s):
    return s.translate(_COMP)[::-1]


class Rec:
    __slots__ = ("r_id", "r_len", "r_al_len", "r_st", "r_en", "r_al", "motif", "strand",
                 "m_al_len", "score", "m_al")


def _split_ws(line, n_fields):
    """Split off n_fields whitespace-separated leading tokens; return tokens + rest."""
    return line.split(None, n_fields)


def parse_report(path, min_record_len=5000):
    """-> (records: OrderedDict r_id -> Rec (oriented), all_alns: dict r_id -> sorted list of
    (r_st, r_en, strand) in file coordinates, read_lens: dict)."""
    with open(path) as fh:
        lines = [ln.strip() for ln in fh]
    lines = [ln for ln in lines if ln and ln[0] != "#"]
    records = OrderedDict()
    all_alns = {}
    read_lens = {}
    for i in range(0, len(lines), 2):
        a, b = lines[i], lines[i + 1]
        r_id, r_len, alen, span, r_al = _split_ws(a, 4)
        assert alen.endswith("bp")
        r_len, alen = int(r_len), int(alen[:-2])
        st, en = (int(x) for x in span.split("-"))
        # second line: motif immediately followed by strand
        cut = min(x for x in (b.find("+"), b.find("-")) if x >= 0)
        motif, strand = b[:cut], b[cut]
        mlen, score, m_al = _split_ws(b[cut + 1:], 2)
        all_alns.setdefault(r_id, []).append((st, en, strand))
        read_lens[r_id] = r_len
        cur = records.get(r_id)
        if cur is None or cur.r_al_len < alen:
            if alen < min_record_len:
                continue
            if strand == "-":
                st, en = r_len - en, r_len - st
                r_al, m_al = revcomp(r_al), revcomp(m_al)
            rec = Rec()
            rec.r_id, rec.r_len, rec.r_al_len, rec.r_st, rec.r_en = r_id, r_len, alen, st, en
            rec.r_al, rec.m_al, rec.motif, rec.strand = r_al, m_al, motif, strand
            rec.m_al_len, rec.score = int(mlen[:-2]), int(score.split("=")[1])
            records[r_id] = rec  # keeps first-insertion position on replacement
    for v in all_alns.values():
        v.sort()
    return records, all_alns, read_lens


def unit_columns(rec, n=1):
    """Unit boundaries [c0, c1, ..., cm] in alignment columns (units are [c_t, c_t+1))."""
    m_al = rec.m_al.upper()
    cols = [c for c, ch in enumerate(m_al) if ch != "-"]
    flat = m_al.replace("-", "")
    pat = rec.motif * n
    starts, pos, last_end = [], 0, None
    while True:
        p = flat.find(pat, pos)
        if p < 0:
            break
        end = cols[p + len(pat) - 1] + 1
        while end < len(m_al) and m_al[end] == "-":
            end += 1
        starts.append(cols[p])
        last_end = end
        pos = p + len(pat)
    if not starts:
        return []
    coords = starts + [last_end]
    if coords[0] > len(rec.motif) * 0.2:
        coords.insert(0, 0)
    if coords[-1] < len(rec.r_al) - len(rec.motif) * 0.2:
        coords.append(len(rec.r_al))
    return coords


def unit_reads(rec, n=1):
    """De-gapped, upper-cased read sequence of each unit (read_kmer_cloud.py:23-25)."""
    c = unit_columns(rec, n)
    return [rec.r_al[a:b].upper().replace("-", "") for a, b in zip(c[:-1], c[1:])]


def classify(records, all_alns, read_lens, large_threshold, small_threshold=1000):
    """-> (prefix ids, internal ids, suffix ids) in record order."""
    pre, mid, suf = [], [], []
    for r_id, rec in records.items():
        r_len = read_lens[r_id]
        alns = all_alns[r_id]
        if rec.strand == "+":
            left, right = alns[0][0], alns[-1][1]
        else:
            left, right = r_len - alns[-1][1], r_len - alns[0][0]
        if left > large_threshold and right > r_len - small_threshold and right == rec.r_en:
            pre.append(r_id)
        elif right < r_len - large_threshold and left < small_threshold and left == rec.r_st:
            suf.append(r_id)
        else:
            mid.append(r_id)
    return pre, mid, suf
