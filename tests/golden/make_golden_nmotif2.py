#!/usr/bin/env python3
"""Stage 3 with --n-motif 2, captured from the REFERENCE ITSELF (build container only: imports /root/reference/scripts read-only,
like make_golden.py).  Units are then PAIRS of motif copies (ncrf_parser.py:28-59 with motif * n, read_placer.py:141-142); the
reference's ReadPlacer runs on the fixture's report with the unique k-mers the reference selected in stage 2 (the committed
tests/golden/<name>.unique_kmers.txt) and writes read_positions.csv.  Golden content per fixture -> tests/golden/<name>.n_motif2.json:
unit columns of EVERY kept record for n = 2, digests of the clouds before / after the multiplicity filter, the placed lines in order
and the None lines sorted, the contig's extent.

    PYTHONHASHSEED=1 python tests/golden/make_golden_nmotif2.py [fixture ...]
    PYTHONHASHSEED=2 python tests/golden/make_golden_nmotif2.py --check [fixture ...]   # must agree
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import canon  # noqa: E402
import fixtures  # noqa: E402
from make_golden import import_reference  # noqa: E402


def capture(name, workdir):
    ncrf_parser, D, RP, RKC, CC = import_reference()
    report = fixtures.make_report(name, workdir)
    p3 = dict(fixtures.stage3_params(name), n_motif=2)
    kfile = os.path.join(HERE, f"{name}.unique_kmers.txt")
    out3 = os.path.join(workdir, f"{name}_stage3_n2")
    params = types.SimpleNamespace(ncrf=report, genomic_kmers=kfile, outdir=out3, n_motif=2, k_cloud=p3["k_cloud"], min_cloud_kmer_freq=p3["min_cloud_kmer_freq"],
                                   min_kmer_mult=p3["min_kmer_mult"], min_unit=p3["min_unit"], min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    g = dict(fixture=name, report_sha256=fixtures.sha256_file(report), stage3=p3, unique_kmers_sha256=fixtures.sha256_file(kfile))
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        placer = RP.ReadPlacer(params)
        rep = placer.ncrf_report
        g["unit_cols_n2"] = {}
        for r_id, rec in rep.records.items():
            mas = rec.get_motif_alignments(n=2)
            g["unit_cols_n2"][r_id] = [ma.start for ma in mas] + ([mas[-1].end] if mas else [])
        c3 = RKC.get_reads_kmer_clouds(rep, n=2, k=p3["k_cloud"], genomic_kmers=placer.genomic_kmers)
        cl3 = [[sorted(c) for c in c3[r_id].kmers] for r_id in rep.records]
        g["clouds3"] = dict(sizes=[[len(c) for c in units] for units in cl3], digest=canon.clouds_digest(cl3))
        c3f = RKC.filter_reads_kmer_clouds(c3, min_mult=p3["min_kmer_mult"])
        cl3f = [[sorted(c) for c in c3f[r_id].kmers] for r_id in rep.records]
        g["clouds3_filtered"] = dict(sizes=[[len(c) for c in units] for units in cl3f], digest=canon.clouds_digest(cl3f))
        placer.run()
        cc = placer.cloud_contig
        g["contig"] = dict(max_pos=cc.max_pos, n_freq_kmers=len(cc.freq_kmers), coverage=sorted(cc.coverage.items()))
    with open(os.path.join(out3, "read_positions.csv")) as f:
        lines = [ln.rstrip("\n") for ln in f]
    g["read_positions"] = dict(placed=[ln for ln in lines if not ln.endswith(" None")], none=sorted(ln for ln in lines if ln.endswith(" None")))
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(fixtures.FIXTURES))
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        for name in a.names:
            g = capture(name, wd)
            path = os.path.join(HERE, f"{name}.n_motif2.json")
            if a.check:
                with open(path) as f:
                    same = json.dumps(json.load(f), sort_keys=True) == json.dumps(g, sort_keys=True)
                print(f"{name}: {'IDENTICAL' if same else 'DIFFERENT'} under PYTHONHASHSEED={os.environ.get('PYTHONHASHSEED', '')}")
                if not same:
                    sys.exit(1)
            else:
                with open(path, "w") as f:
                    json.dump(g, f, indent=0, sort_keys=True)
                print(f"{name}: wrote {path} (placed={len(g['read_positions']['placed'])} none={len(g['read_positions']['none'])} max_pos={g['contig']['max_pos']})")


if __name__ == "__main__":
    main()
