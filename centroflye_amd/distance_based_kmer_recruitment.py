"""Drop-in counterpart of the reference's ``scripts/distance_based_kmer_recruitment.py`` (stage 2).

Same function names, argument meaning, CLI flags and output files; the work runs on the GPU
through the C ABI (include/cfhip.h).  Where the reference returns dicts / sets of Python strings
this module returns array-backed objects with the same mapping / set protocol
(centroflye_amd/kmers.py).  The (a, b, d) histogram of ``get_kmer_dist_map`` (reference :85-128)
is never materialised — the reference needs up to 800 GB for it (README.md:75) — the returned
handle carries the request and ``filter_dist_tuples`` runs the fused histogram + filter kernel.

Extra CLI flags (not in the reference): ``--no-edges`` skips the edge text file that nothing
downstream reads (the reference only writes it, :165-171), ``--metrics`` writes a JSON side file.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

from . import _host
from . import kmers as km
from . import session
from .ncrf_parser import NCRF_Report
from .read_kmer_cloud import get_reads_kmer_clouds

EDGE_CHUNK = 1 << 25   # edges fetched from the GPU per partition pass (512 MiB)


def smart_makedirs(dirname):
    os.makedirs(dirname, exist_ok=True)   # reference: utils/os_utils.py:29-34


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--ncrf", required=True, help="NCRF report on reads")
    p.add_argument("--coverage", type=int, required=True, help="Average coverage of the dataset")
    p.add_argument("--min-coverage", type=int, default=4, help="minCov threshold")
    p.add_argument("--outdir", required=True, help="Output directory")
    p.add_argument("-k", type=int, default=19)
    p.add_argument("--min-nreads", type=int, default=0)
    p.add_argument("--max-nreads", type=int, default=sys.maxsize)
    p.add_argument("--min-distance", type=int, default=1)
    p.add_argument("--max-distance", type=int, default=150)
    p.add_argument("--bottom", type=float, default=0.9)
    p.add_argument("--top", type=float, default=3.0)
    p.add_argument("--kmer-survival-rate", type=float, default=0.34)
    p.add_argument("--max-nonuniq", type=int, default=3)
    p.add_argument("--verbose", action="store_true", default=True)
    p.add_argument("--no-edges", action="store_true", help="do not write unique_edges_min_edge_cov_N.txt")
    p.add_argument("--metrics", action="store_true", help="write stage2_metrics.json beside the outputs")
    return p.parse_args(argv)


def rare_window(bottom, top, coverage, kmer_survival_rate):
    """Integer [lo, hi] with lo <= f <= hi  <=>  left <= f <= right for the reference's own double
    expressions (reference :74-78)."""
    left = bottom * coverage * kmer_survival_rate
    right = top * coverage * kmer_survival_rate
    lo = max(0, int(left))
    while lo < left:
        lo += 1
    hi = int(right) + 1
    while hi > right:
        hi -= 1
    return lo, hi


def get_kmer_freqs_from_ncrf_report(reads_ncrf_report, k, verbose, max_nonuniq):
    """{k-mer: number of reads holding it} without the k-mers that occur twice in more than max_nonuniq reads (reference :39-63).
    The reference's dict also holds the windows with a symbol other than upper-case A, C, G, T as string keys of their own (it counts
    the raw text): those come from the host side path (cfh_exotic_kept) and ride along as a plain dict (KmerFreqs.extra)."""
    packed = reads_ncrf_report.packed
    extra = packed.exotic_kept(k, max_nonuniq) if packed.non_acgt else None
    e = session.ensure_loaded(packed, 1)
    e.count_kmers(k)
    keys, pres, multi = e.table()
    keep = multi.astype(np.int64) <= max_nonuniq
    return km.KmerFreqs(keys[keep], pres[keep], k, extra)


def check_exotic_windows(packed, k, max_nonuniq, lo, hi, verbose=False):
    """Reads with symbols other than upper-case A, C, G, T (N calls, soft-masked stretches): the reference counts the
    windows that hold one as k-mers of their own (reference :47-53, no upper-casing there) — the device path skips them.
    That changes an output only if such a k-mer is rare AND can match a window of an upper-cased unit
    (read_kmer_cloud.py:25), i.e. holds no lower-case letter (the summary's n_blocking): get_rare_kmers then carries those
    k-mers as strings beside the 2-bit set (kmers.KmerSet.extra).  Returns the summary, or None for plain reads."""
    if not packed.non_acgt:
        return None
    ex = packed.exotic_summary(k, max_nonuniq, lo, hi)
    if verbose:
        print(f"# k-mers with symbols other than ACGT: {ex['n_distinct']} (rare: {ex['n_rare']}, of them without a lower-case letter: {ex['n_blocking']})")
    return ex


def get_rare_kmers(reads_ncrf_report, k, bottom, top, coverage, kmer_survival_rate, max_nonuniq, verbose):
    packed = reads_ncrf_report.packed
    e = session.ensure_loaded(packed, 1)
    e.count_kmers(k)
    lo, hi = rare_window(bottom, top, coverage, kmer_survival_rate)
    ex = check_exotic_windows(packed, k, max_nonuniq, lo, hi, verbose)
    e.select_rare(max_nonuniq, lo, hi)
    extra = packed.exotic_rare(k, max_nonuniq, lo, hi) if ex and ex["n_blocking"] else []
    # (the rare windows WITH a lower-case letter: the reference's set holds them too — members only, no cloud can)
    inert = packed.exotic_rare_lower(k, max_nonuniq, lo, hi) if ex and ex["n_rare"] > ex["n_blocking"] else []
    rare = km.KmerSet(e.kmers(), k, extra, inert)
    if extra:       # the device's list gets their pseudo-codes: ranks, posting lists and edges like any other k-mer's
        e.set_kmers(rare.codes, k)
    if verbose:
        print(f"# rare kmers: {len(rare)}")
    return rare


class DistMap:
    """Handle standing for the reference's ``dist_cnt`` (histogram over (d, a, b))."""

    def __init__(self, clouds, min_n, max_n, min_d, max_d):
        self.clouds, self.min_n, self.max_n, self.min_d, self.max_d = clouds, min_n, max_n, min_d, max_d
        self.stats = None


def get_kmer_dist_map(reads_kmer_clouds, kmers, min_n, max_n, min_d, max_d, verbose):
    kset = reads_kmer_clouds.kset
    if len(kmers) != len(kset):
        raise ValueError("the k-mer set must be the one the clouds were built from")
    return DistMap(reads_kmer_clouds, min_n, max_n, min_d, max_d), km.KmerIndex(kset)


def _run_dist(dist_cnt, min_coverage, rel_threshold, edge_sink=None):
    """Run the fused kernel over first-k-mer partitions small enough to fetch; returns the unique
    mask.  edge_sink(edges uint32[n,4]) is called per partition (edges sorted by (d, a, b))."""
    e = dist_cnt.clouds.on_device()
    n_parts = 1
    chunk = EDGE_CHUNK
    n_first = max(1, int(dist_cnt.clouds.kset.codes.size))      # (more partitions than first k-mers split nothing: then the chunk grows)
    while True:
        e.reset_unique()
        ok = True
        total = 0
        for part in range(n_parts):
            n = e.dist_edges(dist_cnt.min_n, dist_cnt.max_n, dist_cnt.min_d, dist_cnt.max_d, min_coverage, rel_threshold,
                             part, n_parts, chunk if edge_sink is not None else 0)
            total += n
            if edge_sink is not None:
                if n > chunk:      # partition too large to fetch: start over with more partitions
                    if n_parts >= n_first:
                        chunk = int(n)
                    else:
                        n_parts = min(max(2 * n_parts, int(n_parts * (n / chunk) * 1.5) + 1), n_first)
                    edge_sink(None)
                    ok = False
                    break
                e.sort_edges()          # by (d, a, b), on the device
                edge_sink(e.edges(n))
        if ok:
            break
    dist_cnt.stats = e.stats()
    dist_cnt.stats["n_edges"] = total
    return e.unique_mask()


class EdgeList(list):
    """list of (d, i, j, freq) tuples; ``.array`` keeps the uint32[n, 4] form."""
    array = None


def filter_dist_tuples(dist_cnt, min_coverage, rel_threshold=0.8):
    parts = []
    mask = _run_dist(dist_cnt, min_coverage, rel_threshold, lambda ed: parts.clear() if ed is None else parts.append(ed))
    arr = np.concatenate(parts) if parts else np.zeros((0, 4), np.uint32)
    edges = EdgeList(map(tuple, arr.tolist()))
    edges.array = arr
    return set(np.flatnonzero(mask).tolist()), edges


def write_kmer_file(path, kset, ranks):
    """The k-mers of the given ranks, one per line, sorted as strings (reference :157-162)."""
    ranks = np.asarray(ranks, np.int64)
    plain = ranks[ranks < kset.n_acgt]
    _host.write_kmers(path, kset.codes[np.sort(plain)], kset.k)     # ascending codes = sorted strings
    if plain.size < ranks.size:       # k-mers without a 2-bit code: merged into the sorted lines
        import heapq
        more = sorted(kset.strings(ranks[ranks >= kset.n_acgt]))
        with open(path) as f:
            lines = f.read().splitlines()
        with open(path, "w") as f:
            f.write("".join(ln + "\n" for ln in heapq.merge(lines, more)))


def write_edge_file(path, kset, edges, append=False):
    """Rows (d, a, b, freq) as "d kmer_a kmer_b freq" lines (reference :165-170)."""
    edges = np.asarray(edges, np.uint32).reshape(-1, 4)
    odd = (edges[:, 1] >= kset.n_acgt) | (edges[:, 2] >= kset.n_acgt) if kset.extra else None
    if odd is None or not odd.any():
        _host.write_edges(path, kset.codes, kset.k, edges, append=append)
        return
    _host.write_edges(path, kset.codes, kset.k, np.ascontiguousarray(edges[~odd]), append=append)
    rows = edges[odd]
    a, b = kset.strings(rows[:, 1]), kset.strings(rows[:, 2])
    with open(path, "a") as f:
        f.write("".join(f"{int(r[0])} {x} {y} {int(r[3])}\n" for r, x, y in zip(rows, a, b)))


def output_results(kmer_index, min_coverage, unique_kmers_ind, dist_edges, outdir):
    kset = kmer_index.kset
    kfile = os.path.join(outdir, f"unique_kmers_min_edge_cov_{min_coverage}.txt")
    write_kmer_file(kfile + ".tmp", kset, np.array(sorted(unique_kmers_ind), np.int64))
    os.replace(kfile + ".tmp", kfile)
    efile = os.path.join(outdir, f"unique_edges_min_edge_cov_{min_coverage}.txt")
    arr = dist_edges.array if getattr(dist_edges, "array", None) is not None else np.array(list(dist_edges), np.uint32).reshape(-1, 4)
    write_edge_file(efile + ".tmp", kset, arr)
    os.replace(efile + ".tmp", efile)


def main(argv=None):
    # N GPUs behind the same command line (centroflye_amd/sharded_cli.py): CF_GPUS=N starts N ranks from this process — which then
    # never touches a GPU —; under a launcher (RANK / WORLD_SIZE > 1 in the environment, e.g. torch.distributed.run) or with
    # CF_SHARDED=1 this process IS a rank
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if (world > 1 and "RANK" in os.environ) or os.environ.get("CF_SHARDED") == "1":
        from . import sharded_cli
        return sharded_cli.rank_main(argv)
    n_gpus = int(os.environ.get("CF_GPUS", "0") or 0)
    if n_gpus > 1:
        from . import sharded_cli
        rc = sharded_cli.launch(argv, n_gpus)
        if rc:
            raise SystemExit(rc)
        return 0
    params = parse_args(argv)
    smart_makedirs(params.outdir)
    t0 = time.time()
    reads_ncrf_report = NCRF_Report(params.ncrf, keep_rows=False)
    t_parse = time.time() - t0
    rare_kmers = get_rare_kmers(reads_ncrf_report, k=params.k, bottom=params.bottom, top=params.top, coverage=params.coverage,
                                kmer_survival_rate=params.kmer_survival_rate, max_nonuniq=params.max_nonuniq, verbose=params.verbose)
    reads_kmer_clouds = get_reads_kmer_clouds(reads_ncrf_report, n=1, k=params.k, genomic_kmers=rare_kmers)
    dist_cnt, kmer_index = get_kmer_dist_map(reads_kmer_clouds, rare_kmers, min_n=params.min_nreads, max_n=params.max_nreads,
                                             min_d=params.min_distance, max_d=params.max_distance, verbose=params.verbose)
    # fused fast path: stream the edges partition by partition straight into the text file
    kset = kmer_index.kset
    efile = os.path.join(params.outdir, f"unique_edges_min_edge_cov_{params.min_coverage}.txt")
    sink = None
    if not params.no_edges:
        open(efile + ".tmp", "w").close()

        def sink(ed):
            if ed is None:
                open(efile + ".tmp", "w").close()
            else:
                write_edge_file(efile + ".tmp", kset, ed, append=True)
    t1 = time.time()
    mask = _run_dist(dist_cnt, params.min_coverage, 0.8, sink)
    t_dist = time.time() - t1
    if not params.no_edges:
        os.replace(efile + ".tmp", efile)
    kfile = os.path.join(params.outdir, f"unique_kmers_min_edge_cov_{params.min_coverage}.txt")
    write_kmer_file(kfile + ".tmp", kset, np.flatnonzero(mask))
    os.replace(kfile + ".tmp", kfile)
    if params.verbose:
        print(f"# unique kmers: {int(mask.sum())}; edges: {dist_cnt.stats['n_edges']}")
    if params.metrics:
        e = session.engine()
        with open(os.path.join(params.outdir, "stage2_metrics.json"), "w") as f:
            json.dump(dict(stats=dist_cnt.stats, times_ms=e.times(), parse_s=t_parse, dist_wall_s=t_dist,
                           total_s=time.time() - t0, device=e.device_info()), f, indent=1)


if __name__ == "__main__":
    main()
