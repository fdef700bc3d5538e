"""Drop-in counterpart of the reference's ``scripts/ncrf_parser.py`` (same class and method names).

The parsing, orientation, unit split and classification run in the native host library
(``libcfhost.so``, include/cfhost.h) and land in flat arrays (``NCRF_Report.packed``) that the
device pipeline consumes; the record objects below are thin views over those arrays so that
code written against the reference (``report.records[r_id].r_al`` ...) keeps working.

Reference: scripts/ncrf_parser.py:11-59 (NCRF_Record), :61-118 (__init__), :120-145 (classify),
:174-178 (get_motif_alignments).
"""
from collections import namedtuple

from . import _host

MotifAlignment = namedtuple("MotifAlignment", ["r_id", "start", "end", "r_al", "m_al"])


class NCRF_Report:
    class NCRF_Record:
        """One kept alignment (already oriented: '-' records are reverse-complemented)."""

        def __init__(self, report, index):
            self._report = report
            self._index = index
            m = report.packed.meta[index]
            self.r_id = report.packed.ids[index]
            self.r_len, self.r_al_len, self.r_st, self.r_en = int(m[0]), int(m[1]), int(m[2]), int(m[3])
            self.strand = "+-"[int(m[4])]
            self.motif = report.packed.motifs[int(m[7])]

        @property
        def r_al(self):
            return self._report.packed.row(self._index, 0)

        @property
        def m_al(self):
            return self._report.packed.row(self._index, 1)

        def get_motif_alignments(self, n=1, overlapped=False):
            if overlapped:
                raise NotImplementedError("overlapped matches are not used by the pipeline")
            up, _, _, uc = self._report.packed.units(n)
            r_al, m_al = self.r_al, self.m_al
            return [MotifAlignment(self.r_id, int(a), int(b), r_al[a:b], m_al[a:b])
                    for a, b in uc[up[self._index]:up[self._index + 1]]]

    def __init__(self, report_fn, min_record_len=5000, keep_rows=True):
        self.packed = _host.parse_report(report_fn, min_record_len=min_record_len, keep_rows=keep_rows)
        self.records = {r_id: self.NCRF_Record(self, i) for i, r_id in enumerate(self.packed.ids)}
        self.discarded_reads = list(self.packed.discarded_reads)
        self.read_lens = {rec.r_id: rec.r_len for rec in self.records.values()}

    def classify(self, large_threshold, small_threshold=1000):
        cls = self.packed.classify(large_threshold, small_threshold)
        ids = self.packed.ids
        pick = lambda c: [ids[i] for i in range(len(ids)) if cls[i] == c]
        return pick(0), pick(1), pick(2)   # prefix, internal, suffix (the reference's return order)

    def get_motif_alignments(self, n=1):
        return {r_id: rec.get_motif_alignments(n=n) for r_id, rec in self.records.items()}
