#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER ONLY: needs /root/reference, no GPU): randomised differential test of the IMPORTABLE stage-2 / cloud API of
SURVEY §8(b) — the functions a script may call without going through the command lines (which take a fused path of their own) — on the
host-emulated kernels against the reference's own functions, in one process:
  get_kmer_freqs_from_ncrf_report (the whole {k-mer: presence} mapping), get_rare_kmers (the set), get_reads_kmer_clouds (per read: r_id, the
  list of per-unit sets, all_kmers; n = 1 or 2), filter_reads_kmer_clouds (min_mult / max_mult), get_kmer_dist_map + filter_dist_tuples
  (unique k-mers and edges, index numbering undone through kmer_index; other rel_threshold values), output_results (both files).
usage: tools/fuzz_api2_vs_reference.py [cases] [--seed S] [--seconds T]"""
import json, os, random, sys, tempfile, time, types, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/scripts"


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    if not os.path.isdir(REF):
        sys.exit("this tool needs /root/reference (build container only)")
    import subprocess
    sys.dont_write_bytecode = True
    bio = types.ModuleType("Bio"); bio.SeqIO = types.ModuleType("Bio.SeqIO"); sys.modules["Bio"] = bio; sys.modules["Bio.SeqIO"] = bio.SeqIO
    sys.path.insert(0, REF)
    import ncrf_parser as RNP, distance_based_kmer_recruitment as RD, read_kmer_cloud as RK      # the reference's (flat names)
    import fixtures
    from centroflye_amd import _host, _lib, session
    from centroflye_amd import distance_based_kmer_recruitment as OD, read_kmer_cloud as OK
    from centroflye_amd.engine import Engine
    from centroflye_amd.ncrf_parser import NCRF_Report as ONR
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    session.reset()
    session._engine = Engine(0, _lib.load(os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so")))
    session._engine.set_param("dist_slots", 2048); session._engine.set_param("dist_block", 128)
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
    seed, budget = arg("--seed", 1), arg("--seconds", 10 ** 9, float)
    rng = random.Random(seed)
    recs, t_start = [], time.time()
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        unit_len = rng.choice([120, 200, 342, 684])
        sy = dict(seed=rng.randrange(1, 1 << 30), unit_len=unit_len, monomer_len=unit_len // rng.choice([2, 3, 4]), n_units=rng.choice([20, 40]), flank=20000, n_reads=rng.choice([10, 16, 24]),
                  mean_len=float(max(5200, unit_len * rng.choice([4, 10, 25]))), sigma=0.2, min_len=5100, max_len=20000, unit_div=rng.choice([0.01, 0.03]), p_split=rng.choice([0.0, 0.2]),
                  var_len=rng.choice([1, 8]), p_sub=0.01, p_del=0.01, p_ins=0.01)
        mut = dict(seed=rng.randrange(1, 1000), n_lower=rng.choice([0, 4]), lower_len=rng.choice([5, 40]), n_N=rng.choice([0, 3])) if rng.random() < 0.4 else None
        k = rng.choice([11, 15, 19, 23, 31]); mnu = rng.choice([0, 1, 3]); cov = rng.choice([6, 10, 14]); min_d = rng.choice([1, 2]); max_d = min_d + rng.choice([0, 1, 3])
        min_cov = rng.choice([1, 2, 3, 4]); thr = rng.choice([0.8, 0.8, 0.5, 0.95]); n_motif = rng.choice([1, 2]); min_mult = rng.choice([1, 2, 3]); max_mult = rng.choice([float("inf"), 6])
        lo_r = rng.choice([0, 0, 2]); hi_r = rng.choice([sys.maxsize, sys.maxsize, lo_r + 5])
        rec = dict(case=i, synth=sy, mutate=mut, k=k, max_nonuniq=mnu, coverage=cov, distances=[min_d, max_d], min_coverage=min_cov, rel_threshold=thr, n_motif=n_motif,
                   mult=[min_mult, None if max_mult == float("inf") else max_mult], reads_window=[lo_r, None if hi_r == sys.maxsize else hi_r])
        diffs = []
        t0 = time.time()
        with tempfile.TemporaryDirectory() as work:
            report = os.path.join(work, "report.ncrf")
            try:
                _host.synth(report_path=report, pack=False, **sy)
            except _host.HostError:
                continue
            if mut:
                fixtures.mutate_report(report, **mut)
            try:
                sink = io.StringIO()
                with contextlib.redirect_stdout(sink):
                    rr, orp = RNP.NCRF_Report(report), ONR(report)
                    fr, fo = RD.get_kmer_freqs_from_ncrf_report(rr, k, False, mnu), OD.get_kmer_freqs_from_ncrf_report(orp, k, False, mnu)
                    if dict(fr) != {x: fo[x] for x in fo} or len(fr) != len(fo):
                        diffs.append("k-mer frequencies")
                    ar, ao = RD.get_rare_kmers(rr, k, 0.9, 3.0, cov, 0.34, mnu, False), OD.get_rare_kmers(orp, k, 0.9, 3.0, cov, 0.34, mnu, False)
                    if set(ar) != set(ao):
                        diffs.append("rare k-mers")
                    for n in (1, n_motif):
                        cr, co = RK.get_reads_kmer_clouds(rr, n=n, k=k, genomic_kmers=ar), OK.get_reads_kmer_clouds(orp, n=n, k=k, genomic_kmers=ao)
                        if list(cr) != list(co) or any(cr[x].r_id != co[x].r_id or [set(u) for u in cr[x].kmers] != [set(u) for u in co[x].kmers] for x in cr):
                            diffs.append(f"clouds n={n}")
                        frr, foo = RK.filter_reads_kmer_clouds(cr, min_mult=min_mult, max_mult=max_mult), OK.filter_reads_kmer_clouds(co, min_mult=min_mult, max_mult=max_mult)
                        if list(frr) != list(foo) or any([set(u) for u in frr[x].kmers] != [set(u) for u in foo[x].kmers] for x in frr):
                            diffs.append(f"filtered clouds n={n}")
                    cr, co = RK.get_reads_kmer_clouds(rr, n=1, k=k, genomic_kmers=ar), OK.get_reads_kmer_clouds(orp, n=1, k=k, genomic_kmers=ao)
                    dr, ir = RD.get_kmer_dist_map(cr, ar, lo_r, hi_r, min_d, max_d, False)
                    do, io_ = OD.get_kmer_dist_map(co, ao, lo_r, hi_r, min_d, max_d, False)
                    if set(ir) != set(io_):
                        diffs.append("kmer_index keys")
                    ur, er = RD.filter_dist_tuples(dr, min_cov, thr) if thr != 0.8 else RD.filter_dist_tuples(dr, min_cov)
                    uo, eo = OD.filter_dist_tuples(do, min_cov, thr) if thr != 0.8 else OD.filter_dist_tuples(do, min_cov)
                    rev_r, rev_o = {v: x for x, v in ir.items()}, {v: x for x, v in io_.items()}
                    if {rev_r[x] for x in ur} != {rev_o[x] for x in uo}:
                        diffs.append("unique k-mers")
                    if sorted((d, rev_r[a], rev_r[b], c) for d, a, b, c in er) != sorted((d, rev_o[a], rev_o[b], c) for d, a, b, c in eo):
                        diffs.append("edges")
                    o1, o2 = os.path.join(work, "r"), os.path.join(work, "o")
                    os.makedirs(o1); os.makedirs(o2)
                    RD.output_results(ir, min_cov, ur, er, o1); OD.output_results(io_, min_cov, uo, eo, o2)
                    for fn in (f"unique_kmers_min_edge_cov_{min_cov}.txt", f"unique_edges_min_edge_cov_{min_cov}.txt"):
                        a, b = open(os.path.join(o1, fn)).read().splitlines(), open(os.path.join(o2, fn)).read().splitlines()
                        if (a != b) if fn.startswith("unique_kmers") else (sorted(a) != sorted(b)):
                            diffs.append(fn)
                rec.update(n_freqs=len(fr), n_rare=len(ar), n_unique=len(ur), n_edges=len(er))
            except Exception as ex:
                diffs.append("exception: " + repr(ex)[:300])
        rec.update(identical=not diffs, differences=diffs, s=round(time.time() - t0, 1))
        recs.append(rec)
        print(json.dumps({k_: rec.get(k_) for k_ in ("case", "identical", "differences", "k", "n_motif", "rel_threshold", "reads_window", "n_freqs", "n_rare", "n_unique", "n_edges", "s")}), flush=True)
    bad = [r for r in recs if not r["identical"]]
    print(json.dumps(dict(seed=seed, cases=len(recs), different=len(bad), seconds=round(time.time() - t_start, 1))))
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_api2_vs_reference.json"), str)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(dict(cases=recs), open(out, "w"), indent=1)
    sys.exit(1 if bad else 0)


main()
