"""TEST INFRASTRUCTURE — CPU restatement of the read-unit export of the reference's polisher
(scripts/eltr_polisher.py:19-30 read_reported_positions, :45-51 default max_pos, :53-66 map_pos2read, :68-97
export_read_units; FASTA text as utils/bio.py:32-36 write_bio_seqs prints it).  Only tests/ may import this.
Pinned by tests/golden/*.read_units.json (captured by running the reference itself, tests/golden/make_golden_polisher.py).
"""
import math
import statistics

from . import ncrf


def read_reported_positions(path):
    pos = {}
    with open(path) as f:
        for line in f:
            fields = line.strip().split(' ')            # :24
            pos[fields[0]] = None if fields[1] == 'None' else int(fields[1])   # :25-29
    return pos


def export(records, placement, min_pos=0, max_pos=math.inf):
    """-> {position: (read_units.fasta text, median_read_unit.fasta text)}"""
    units = {}
    for r_id, rec in records.items():
        b = ncrf.unit_columns(rec, 1)
        units[r_id] = list(zip(b[:-1], b[1:]))
    if max_pos == math.inf:                             # :45-51
        max_pos = 0
        for r_id, pos in placement.items():
            if pos is not None:
                max_pos = max(max_pos, pos + len(units[r_id]))
    pos2read = {}
    for r_id, pos in placement.items():                 # :55-65
        if pos is None or pos > max_pos:
            continue
        n = len(units[r_id])
        idx = range(n) if (pos == min_pos or pos + n == max_pos) else range(1, n - 1)
        for i in idx:
            if min_pos <= pos + i <= max_pos:
                pos2read.setdefault(pos + i, []).append((r_id, i))
    out = {}
    for pos, members in pos2read.items():               # :70-96
        seqs = {}
        for r_id, p in members:
            a, b = units[r_id][p]
            seqs[f'gen_pos={pos}|r_id={r_id}|r_pos={p}'] = records[r_id].r_al[a:b].upper().replace('-', '')
        med_len = statistics.median_high([len(s) for s in seqs.values()])
        template = next(k for k in sorted(seqs) if len(seqs[k]) == med_len)
        out[pos] = (''.join(f'>{k}\n{s}\n' for k, s in seqs.items()), f'>{template}\n{seqs[template]}\n')
    return out
