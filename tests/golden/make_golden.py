#!/usr/bin/env python3
"""Capture golden vectors by running the REFERENCE ITSELF on the seeded fixtures.

Runs only in the build container (needs /root/reference, which does not exist on the GPU
box).  It imports the reference's modules read-only (stubbing the absent Biopython, which the
hot path never calls — SURVEY.md Appendix D), runs stage 2 and stage 3 on each fixture of
tests/fixtures.py and writes tests/golden/<name>.json (+ the small output files).

    PYTHONHASHSEED=1 python tests/golden/make_golden.py [fixture ...]
    PYTHONHASHSEED=2 python tests/golden/make_golden.py --check [fixture ...]   # must agree

Golden content: G0 record selection / orientation / unit columns / classes,
G1 presence table digest, G2 rare set, G3 clouds (stage 2 and stage 3, before/after the
multiplicity filter), G4 histogram + selected edges (digests), G5 unique_kmers file, G6
read_positions.csv (placed lines in order, None lines sorted).
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import canon  # noqa: E402
import fixtures  # noqa: E402

REF = "/root/reference/scripts"


def import_reference():
    sys.dont_write_bytecode = True
    bio = types.ModuleType("Bio")
    bio.SeqIO = types.ModuleType("Bio.SeqIO")
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqIO"] = bio.SeqIO
    sys.path.insert(0, REF)
    import ncrf_parser
    import distance_based_kmer_recruitment as D
    import read_placer as RP
    import read_kmer_cloud as RKC
    import cloud_contig as CC
    return ncrf_parser, D, RP, RKC, CC


def capture(name, workdir):
    ncrf_parser, D, RP, RKC, CC = import_reference()
    report = fixtures.make_report(name, workdir)
    p2 = fixtures.stage2_params(name)
    p3 = fixtures.stage3_params(name)
    g = dict(fixture=name, report_sha256=fixtures.sha256_file(report), stage2=p2, stage3=p3,
             pythonhashseed=os.environ.get("PYTHONHASHSEED", ""))
    t0 = time.time()
    rep = ncrf_parser.NCRF_Report(report)
    # ---- G0
    recs = []
    for r_id, rec in rep.records.items():
        mas = rec.get_motif_alignments(n=1)
        cols = [ma.start for ma in mas] + ([mas[-1].end] if mas else [])
        recs.append(dict(r_id=r_id, strand=rec.strand, r_len=rec.r_len, r_al_len=rec.r_al_len,
                         r_st=rec.r_st, r_en=rec.r_en,
                         r_al_sha1=hashlib.sha1(rec.r_al.encode()).hexdigest(),
                         m_al_sha1=hashlib.sha1(rec.m_al.encode()).hexdigest(), unit_cols=cols))
    g["records"] = recs
    g["discarded"] = sorted(rep.discarded_reads)
    pre, mid, suf = rep.classify(large_threshold=p3["prefix_threshold"])
    g["classify"] = dict(prefix=pre, internal=mid, suffix=suf)
    # n = 2 unit split on the first 5 records (API parity for n_motif != 1)
    g["unit_cols_n2"] = {}
    for r_id, rec in list(rep.records.items())[:5]:
        mas = rec.get_motif_alignments(n=2)
        g["unit_cols_n2"][r_id] = [ma.start for ma in mas] + ([mas[-1].end] if mas else [])
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        # ---- G1
        freqs = D.get_kmer_freqs_from_ncrf_report(rep, k=p2["k"], verbose=False, max_nonuniq=p2["max_nonuniq"])
        g["presence"] = dict(n=len(freqs), digest=canon.presence_digest(freqs.items()))
        # ---- G2
        rare = D.get_rare_kmers(rep, k=p2["k"], bottom=p2["bottom"], top=p2["top"], coverage=p2["coverage"],
                                kmer_survival_rate=p2["kmer_survival_rate"], max_nonuniq=p2["max_nonuniq"],
                                verbose=False)
        g["rare"] = dict(n=len(rare), digest=canon.set_digest(rare))
        # ---- G3 (stage 2 clouds)
        clouds = RKC.get_reads_kmer_clouds(rep, n=1, k=p2["k"], genomic_kmers=rare)
        cl = [[sorted(c) for c in clouds[r_id].kmers] for r_id in rep.records]
        g["clouds2"] = dict(sizes=[[len(c) for c in units] for units in cl], digest=canon.clouds_digest(cl))
        # ---- G4
        dist_cnt, kmer_index = D.get_kmer_dist_map(clouds, rare, p2["min_nreads"], p2["max_nreads"],
                                                   p2["min_distance"], p2["max_distance"], False)
        rev = {i: kmer for kmer, i in kmer_index.items()}
        E = 0
        n_keys = 0
        h = []
        for d, dt in dist_cnt.items():
            for i, row in enumerate(dt):
                for j, c in row.items():
                    if c > 0:
                        E += c
                        n_keys += 1
                        h.append((rev[i], rev[j], d, c))
        g["hist"] = dict(E=E, n_keys=n_keys, digest=canon.hist_digest(h))
        del h
        uk, edges = D.filter_dist_tuples(dist_cnt, min_coverage=p2["min_coverage"])
        del dist_cnt
        outdir = os.path.join(workdir, f"{name}_stage2")
        os.makedirs(outdir, exist_ok=True)
        D.output_results(kmer_index, p2["min_coverage"], uk, edges, outdir)
    kfile = os.path.join(outdir, f"unique_kmers_min_edge_cov_{p2['min_coverage']}.txt")
    efile = os.path.join(outdir, f"unique_edges_min_edge_cov_{p2['min_coverage']}.txt")
    with open(efile) as f:
        elines = [ln.rstrip("\n") for ln in f]
    g["edges"] = dict(n=len(elines), digest=canon.edge_lines_digest(elines))
    with open(kfile) as f:
        ktext = f.read()
    g["unique_kmers"] = dict(n=ktext.count("\n"), sha256=hashlib.sha256(ktext.encode()).hexdigest())
    # ---- stage 3
    out3 = os.path.join(workdir, f"{name}_stage3")
    params = types.SimpleNamespace(ncrf=report, genomic_kmers=kfile, outdir=out3, n_motif=p3["n_motif"],
                                   k_cloud=p3["k_cloud"], min_cloud_kmer_freq=p3["min_cloud_kmer_freq"],
                                   min_kmer_mult=p3["min_kmer_mult"], min_unit=p3["min_unit"],
                                   min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    with contextlib.redirect_stdout(sink):
        placer = RP.ReadPlacer(params)
        gk = placer.genomic_kmers
        c3 = RKC.get_reads_kmer_clouds(placer.ncrf_report, n=p3["n_motif"], k=p3["k_cloud"], genomic_kmers=gk)
        cl3 = [[sorted(c) for c in c3[r_id].kmers] for r_id in placer.ncrf_report.records]
        g["clouds3"] = dict(digest=canon.clouds_digest(cl3))
        c3f = RKC.filter_reads_kmer_clouds(c3, min_mult=p3["min_kmer_mult"])
        cl3f = [[sorted(c) for c in c3f[r_id].kmers] for r_id in placer.ncrf_report.records]
        g["clouds3_filtered"] = dict(sizes=[[len(c) for c in units] for units in cl3f],
                                     digest=canon.clouds_digest(cl3f))
        placer.run()
        # ---- G7 (A10, dead code in the pipeline; API parity only): slow vs fast scoring of one read
        cc = placer.cloud_contig
        some = [r for r in placer.ncrf_report.records][:8]
        g7 = {}
        for r_id in some:
            sc, pos = cc.calc_inters_score(c3f[r_id], min_unit=2, min_inters=10)
            g7[r_id] = [list(sc), pos]
        g["calc_inters_score"] = g7
        g["contig"] = dict(max_pos=cc.max_pos, n_freq_kmers=len(cc.freq_kmers),
                           coverage=sorted(cc.coverage.items()))
    with open(os.path.join(out3, "read_positions.csv")) as f:
        lines = [ln.rstrip("\n") for ln in f]
    placed = [ln for ln in lines if not ln.endswith(" None")]
    none = sorted(ln for ln in lines if ln.endswith(" None"))
    g["read_positions"] = dict(placed=placed, none=none)
    g["reference_seconds"] = round(time.time() - t0, 2)
    return g, ktext


def strip_volatile(g):
    g = dict(g)
    g.pop("pythonhashseed", None)
    g.pop("reference_seconds", None)
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(fixtures.FIXTURES))
    ap.add_argument("--check", action="store_true", help="recompute and compare with the committed goldens")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as wd:
        for name in a.names:
            g, ktext = capture(name, wd)
            jpath = os.path.join(HERE, f"{name}.json")
            kpath = os.path.join(HERE, f"{name}.unique_kmers.txt")
            if a.check:
                with open(jpath) as f:
                    old = json.load(f)
                same = json.dumps(strip_volatile(old), sort_keys=True) == json.dumps(strip_volatile(g), sort_keys=True)
                with open(kpath) as f:
                    same = same and f.read() == ktext
                print(f"{name}: {'IDENTICAL' if same else 'DIFFERENT'} under PYTHONHASHSEED={g['pythonhashseed']} "
                      f"({g['reference_seconds']} s)")
                if not same:
                    sys.exit(1)
            else:
                with open(jpath, "w") as f:
                    json.dump(g, f, indent=0, sort_keys=True)
                with open(kpath, "w") as f:
                    f.write(ktext)
                print(f"{name}: wrote {jpath} ({g['reference_seconds']} s; rare={g['rare']['n']} "
                      f"E={g['hist']['E']} edges={g['edges']['n']} unique={g['unique_kmers']['n']} "
                      f"placed={len(g['read_positions']['placed'])} none={len(g['read_positions']['none'])})")


if __name__ == "__main__":
    main()
