"""Read-unit export for polishing: the consumer of ``read_positions.csv``.

Mirror of the part of the reference's ``scripts/eltr_polisher.py`` that is pure data movement:
``read_reported_positions`` (:19-30), ``ELTR_Polisher.__init__`` (:33-51, the ``max_pos`` default), ``map_pos2read``
(:53-66) and ``export_read_units`` (:68-97).  The grouping and the FASTA writing run in the compiled host library
(``cfh_export_read_units``: one pass over the packed units, positions written by a thread pool).  What follows in the
reference — Flye runs per position (:99-113), re-reading Flye's output and edlib comparisons (:115-144) — is outside
this repository's scope and is not here: the script ends with the exported files.
"""
import argparse
import math
import os

from . import ncrf_parser


def read_reported_positions(read_positions_fn):
    """{r_id: position or None} in file order (reference :19-30: split on ' ', field 1, 'None' -> None)."""
    pos = {}
    with open(read_positions_fn) as f:
        for line in f:
            fields = line.strip().split(' ')
            pos[fields[0]] = None if fields[1] == 'None' else int(fields[1])
    return pos


class ELTR_Polisher:
    def __init__(self, params):
        self.params = params
        if not os.path.isfile(params.unit):
            raise FileNotFoundError(f"File {params.unit} is not found")
        self.ncrf_report = ncrf_parser.NCRF_Report(params.ncrf)
        os.makedirs(params.outdir, exist_ok=True)
        self.read_placement = read_reported_positions(params.read_placement)
        self.min_pos = params.min_pos
        self.max_pos = params.max_pos
        packed = self.ncrf_report.packed
        self._index = {r_id: i for i, r_id in enumerate(packed.ids)}
        self._n_units = (packed.units(1)[0][1:] - packed.units(1)[0][:-1])
        if self.max_pos == math.inf:
            self.max_pos = 0
            for r_id, pos in self.read_placement.items():
                if pos is not None:
                    self.max_pos = max(self.max_pos, pos + int(self._n_units[self._index[r_id]]))

    def map_pos2read(self):
        """{position: [(r_id, unit index), ...]} — small-object form of what the exporter groups natively."""
        pos2read = {}
        for r_id, pos in self.read_placement.items():
            if pos is None or pos > self.max_pos:
                continue
            n = int(self._n_units[self._index[r_id]])
            units = range(n) if (pos == self.min_pos or pos + n == self.max_pos) else range(1, n - 1)
            for i in units:
                if self.min_pos <= pos + i <= self.max_pos:
                    pos2read.setdefault(pos + i, []).append((r_id, i))
        return pos2read

    def export_read_units(self, pos2read=None):
        """Writes pos_P/read_units.fasta and pos_P/median_read_unit.fasta for every position; returns
        {position: (units_fn, median_read_unit_fn)} like the reference.  ``pos2read`` is accepted for signature
        parity; the files are produced from the placement itself by the native exporter."""
        placed = [(self._index[r_id], pos) for r_id, pos in self.read_placement.items() if pos is not None]
        rec = [x[0] for x in placed]
        pos = [x[1] for x in placed]
        self.ncrf_report.packed.export_read_units(rec, pos, self.params.outdir, self.min_pos, self.max_pos)
        positions = pos2read.keys() if pos2read is not None else self.map_pos2read().keys()
        return {p: (os.path.join(self.params.outdir, f'pos_{p}', 'read_units.fasta'),
                    os.path.join(self.params.outdir, f'pos_{p}', 'median_read_unit.fasta')) for p in positions}

    def run(self, export_only=True):
        return self.export_read_units(self.map_pos2read())


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--read-placement", required=True)
    parser.add_argument("--unit", required=True)
    parser.add_argument("--outdir", required=True)
    parser.add_argument("--ncrf", required=True)
    parser.add_argument("--min-pos", type=int, default=0)
    parser.add_argument("--max-pos", type=int, default=math.inf)
    parser.add_argument("--export-only", action="store_true", help="accepted for compatibility: exporting is all this script does")
    params = parser.parse_args()
    ELTR_Polisher(params).run()


if __name__ == "__main__":
    main()
