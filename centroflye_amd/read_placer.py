"""Drop-in counterpart of the reference's ``scripts/read_placer.py`` (stage 3).

Same CLI flags and the same output file ``read_positions.csv`` (space separated): prefix reads
``r_id 0``, placed reads ``r_id pos s0 s1`` in placement order, unplaced reads ``r_id None``.
``ReadPlacer.run`` keeps the reference's sequence (read_placer.py:96-128) — classify, clouds of the
given genomic k-mers, multiplicity filter, prefix reads at 0, then the greedy stages for internal
and suffix reads — with the clouds, the filter and the whole greedy loop on the GPU
(cf_build_clouds, cf_filter_clouds, cf_place_reads).  The ``None`` lines of a stage are written
in ascending r_id order (the reference's order there is Python set order, i.e. hash-seed dependent).
"""
import argparse
import os

import numpy as np

from . import _host
from . import kmers as km
from .ncrf_parser import NCRF_Report
from .read_kmer_cloud import filter_reads_kmer_clouds, get_reads_kmer_clouds


class ReadPlacer:
    def __init__(self, params):
        self.params = params
        self.ncrf_report = NCRF_Report(params.ncrf, keep_rows=getattr(params, "n_motif", 1) != 1)
        self.k_cloud = params.k_cloud
        if params.genomic_kmers is not None:
            self.genomic_kmers = _host.read_kmers(params.genomic_kmers, params.k_cloud)
            if self.genomic_kmers.size > 1 and not (self.genomic_kmers[1:] > self.genomic_kmers[:-1]).all():      # (stage 2 writes them sorted)
                self.genomic_kmers = np.unique(self.genomic_kmers)
            extra = km.exotic_lines(params.genomic_kmers, params.k_cloud)      # k-mers with an N (...): no 2-bit code, kept as strings
            if extra:
                self.genomic_kmers = km.KmerSet(self.genomic_kmers, params.k_cloud, extra)
        else:
            self.genomic_kmers = None
        os.makedirs(params.outdir, exist_ok=True)
        self.position_outfile = os.path.join(params.outdir, "read_positions.csv")
        self.placements = None

    def run(self):
        p = self.params
        report = self.ncrf_report
        classes = report.packed.classify(p.prefix_threshold)
        print(f"Left: {int((classes == 0).sum())}")
        print(f"FT: {int((classes == 1).sum())}")
        print(f"Right: {int((classes == 2).sum())}")
        clouds = get_reads_kmer_clouds(report, n=p.n_motif, k=p.k_cloud, genomic_kmers=self.genomic_kmers)
        clouds = filter_reads_kmer_clouds(clouds, min_mult=p.min_kmer_mult)
        ids = report.packed.ids
        id_rank = np.argsort(np.argsort(np.array(ids, dtype=object), kind="stable"), kind="stable").astype(np.int32)
        engine = clouds.on_device()
        rd, pos, s0, s1 = engine.place_reads(classes, id_rank, p.min_cloud_kmer_freq, p.min_unit, p.min_inters, 3)
        tmp = self.position_outfile + ".tmp"
        with open(tmp, "w") as f:
            for r, q, a, b in zip(rd.tolist(), pos.tolist(), s0.tolist(), s1.tolist()):
                if a < 0 and q == 0:
                    f.write(f"{ids[r]} 0\n")
                elif q < 0:
                    f.write(f"{ids[r]} None\n")
                else:
                    f.write(f"{ids[r]} {q} {a} {b}\n")
        os.replace(tmp, self.position_outfile)
        self.placements = (rd, pos, s0, s1)
        return self.placements


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--ncrf", required=True, help="NCRF report on reads")
    p.add_argument("--genomic-kmers", required=True, help="Unique genomic kmers if known")
    p.add_argument("--n-motif", type=int, default=1, help="Number of motifs stuck together")
    p.add_argument("--k-cloud", type=int, default=19, help="Size of k-mer for k-mer cloud")
    p.add_argument("--min-cloud-kmer-freq", type=int, default=2, help="Minimal frequency of a kmer in the cloud")
    p.add_argument("--min-kmer-mult", type=int, default=2, help="Minimal frequency of a kmer in input")
    p.add_argument("--min-unit", type=int, default=2, help="Score[0]")
    p.add_argument("--min-inters", type=int, default=10, help="Score[1]")
    p.add_argument("--prefix-threshold", type=int, default=50000, help="Min pre/suffix length for read classification")
    p.add_argument("--outdir", required=True, help="Output directory")
    return p.parse_args(argv)


def main(argv=None):
    ReadPlacer(parse_args(argv)).run()


if __name__ == "__main__":
    main()
