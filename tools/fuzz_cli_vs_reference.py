#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER ONLY: needs /root/reference and no GPU): randomised differential test of the two drop-in command lines
against the reference's own scripts.  Every case writes a small random NCRF report (unit length, units per read, split records, strands,
optionally soft-masked stretches, N calls and N calls shared by several reads), draws the command-line options of both stages
(k, --coverage / --bottom / --top / --kmer-survival-rate, --max-nonuniq, --min-coverage, --min-distance / --max-distance,
--min-nreads / --max-nreads; --n-motif, --min-cloud-kmer-freq, --min-kmer-mult, --min-unit, --min-inters, --prefix-threshold) and runs
  (a) /root/reference/scripts/distance_based_kmer_recruitment.py and read_placer.py themselves (Biopython stubbed: SURVEY App. D), and
  (b) this repo's scripts with the kernels on the host emulator (tests/emu),
each in a process of its own; the k-mer file must be equal byte for byte, the edge file as a sorted list of lines, read_positions.csv in the
order of the placed lines and as a set of None lines.  Small --max-distance: the reference takes 0.5 us per pair emission and the emulator not
much less.  usage: tools/fuzz_cli_vs_reference.py [cases] [--seed S] [--seconds T] [--only i] [--keep DIR]"""
import json, os, random, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/scripts"


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


REF_RUN = r'''
import sys, types
sys.dont_write_bytecode = True
bio = types.ModuleType("Bio"); bio.SeqIO = types.ModuleType("Bio.SeqIO"); sys.modules["Bio"] = bio; sys.modules["Bio.SeqIO"] = bio.SeqIO
sys.path.insert(0, %(ref)r)
import %(module)s as M
sys.argv = [%(module)r] + %(argv)r
M.main()
'''
OUR_RUN = r'''
import sys
sys.path.insert(0, %(root)r)
from centroflye_amd import _lib, session
from centroflye_amd.engine import Engine
session._engine = Engine(0, _lib.load(%(emu)r))
session._engine.set_param("dist_slots", 2048); session._engine.set_param("dist_block", 128)
from centroflye_amd import %(module)s as M
M.main(%(argv)r)
'''
# (d) SURVEY 8(f) rank 3: the per-position read-unit FASTA files the polisher writes before it calls Flye (eltr_polisher.py:53-97), from the
# REFERENCE's read_positions.csv, by the reference's ELTR_Polisher (edlib / Biopython stubbed, the unit FASTA is not used by the export) and by
# this repo's eltr_polisher; --min-pos / --max-pos drawn
POL_REF = r'''
import sys, types, math
sys.dont_write_bytecode = True
for mod in ("Bio", "Bio.SeqIO", "edlib"):
    sys.modules.setdefault(mod, types.ModuleType(mod))
sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
sys.path.insert(0, %(ref)r)
import eltr_polisher as E
E.read_bio_seq = lambda fn: "ACGT"
params = types.SimpleNamespace(unit=%(unit)r, ncrf=%(report)r, outdir=%(out)r, read_placement=%(csv)r, min_pos=%(lo)d, max_pos=%(hi)s)
pol = E.ELTR_Polisher(params)
pol.export_read_units(pol.map_pos2read())
'''
POL_OUR = r'''
import sys
sys.path.insert(0, %(root)r)
from centroflye_amd import eltr_polisher as M
sys.argv = ["eltr_polisher", "--ncrf", %(report)r, "--read-placement", %(csv)r, "--unit", %(unit)r, "--outdir", %(out)r, "--export-only", "--min-pos", str(%(lo)d)] + %(hi_arg)r
M.main()
'''


def digest_tree(outdir):
    import hashlib
    res = {}
    for d in sorted(os.listdir(outdir)) if os.path.isdir(outdir) else []:
        if d.startswith("pos_"):
            res[d] = [hashlib.sha256(open(os.path.join(outdir, d, fn), "rb").read()).hexdigest() for fn in ("read_units.fasta", "median_read_unit.fasta")]
    return res


# (c) in a third of the cases: stage 2 once more as CF_GPUS = 2 or 3 ranks on the emulator (centroflye_amd/sharded_cli.py through tests/sharded_cli_worker.py)
SHARD_RUN = r'''
import os, sys
sys.path.insert(0, %(root)r)
os.environ["OMP_NUM_THREADS"] = "1"; os.environ["CF_TEST_SUB_EDGES"] = %(sub)r; os.environ.pop("CF_PACK_CACHE", None)
from centroflye_amd import sharded_cli
sys.exit(sharded_cli.launch(%(argv)r, %(world)d, rank_cmd=[sys.executable, %(worker)r]))
'''


class TooSlow(Exception):
    pass


def run(code, timeout):
    try:
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, PYTHONHASHSEED=str(random.randrange(1, 10 ** 6))))
    except subprocess.TimeoutExpired:
        raise TooSlow(f"a run took more than {timeout} s")
    return p.returncode, (p.stdout[-1500:] + p.stderr[-3000:])


def main():
    if not os.path.isdir(REF):
        sys.exit("this tool needs /root/reference (build container only)")
    import fixtures
    from centroflye_amd import _host
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    emu = os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so")
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    seed, budget, only, keep = arg("--seed", 1), arg("--seconds", 10 ** 9, float), arg("--only", -1), arg("--keep", "", str)
    rng = random.Random(seed)
    recs, t_start = [], time.time()
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        unit_len = rng.choice([120, 200, 200, 342, 684, 2055])
        upr = rng.choice([3, 6, 12, 25])                      # units per read, about
        sy = dict(seed=rng.randrange(1, 1 << 30), unit_len=unit_len, monomer_len=rng.choice([unit_len // 4, unit_len // 2, 171 if unit_len >= 342 else unit_len // 3]),
                  n_units=rng.choice([20, 40, 80]), flank=rng.choice([20000, 60000]), n_reads=rng.choice([12, 20, 30]),
                  mean_len=float(max(5200, unit_len * upr)), sigma=rng.choice([0.1, 0.3]), min_len=5100, max_len=int(max(6000, 2 * unit_len * upr)),
                  unit_div=rng.choice([0.01, 0.03]), n_prefix=rng.choice([0, 2, 3]), n_suffix=rng.choice([0, 2, 3]), prefix_threshold=rng.choice([5000, 50000]),
                  p_split=rng.choice([0.0, 0.1, 0.3]), var_len=rng.choice([1, 8]), p_sub=rng.choice([0.005, 0.02]), p_del=rng.choice([0.005, 0.02]), p_ins=rng.choice([0.005, 0.015]))
        mut = None
        if rng.random() < 0.5:
            mut = dict(seed=rng.randrange(1, 1000), n_lower=rng.choice([0, 3, 12]), lower_len=rng.choice([5, 60]), n_N=rng.choice([0, 2, 10]))
            if rng.random() < 0.4:
                mut.update(n_shared=rng.choice([1, 3]), shared_lo=rng.choice([3, 5]), shared_hi=rng.choice([8, 14]))
        k = rng.choice([11, 15, 19, 19, 23, 31])
        min_d = rng.choice([1, 1, 2])
        a2 = ["--coverage", str(rng.choice([6, 10, 14, 24])), "--min-coverage", str(rng.choice([1, 2, 3, 4])), "-k", str(k), "--min-distance", str(min_d),
              "--max-distance", str(min_d + rng.choice([0, 1, 2, 4])), "--max-nonuniq", str(rng.choice([0, 1, 3, 5]))]
        if rng.random() < 0.3:
            a2 += ["--bottom", str(rng.choice([0.5, 0.9, 1.3])), "--top", str(rng.choice([2.0, 3.0, 4.5])), "--kmer-survival-rate", str(rng.choice([0.2, 0.34, 0.5]))]
        if rng.random() < 0.3:
            lo_r = rng.choice([0, 2, 5])
            a2 += ["--min-nreads", str(lo_r), "--max-nreads", str(lo_r + rng.choice([1, 6, 100]))]
        a3 = ["--n-motif", str(rng.choice([1, 1, 2])), "--k-cloud", str(k), "--min-cloud-kmer-freq", str(rng.choice([1, 2, 2, 3])), "--min-kmer-mult", str(rng.choice([1, 2, 2, 3])),
              "--min-unit", str(rng.choice([1, 2, 2, 3])), "--min-inters", str(rng.choice([1, 4, 10, 10, 30])), "--prefix-threshold", str(sy["prefix_threshold"])]
        sharded = (rng.choice([2, 3]), rng.choice([0, 0, 20000])) if rng.random() < 0.33 else None      # (ranks, edge rows per sub-partition)
        polish = (rng.choice([0, 0, 1, 3]), rng.choice([None, None, 4, 9])) if rng.random() < 0.6 else None      # (--min-pos, --max-pos) of the export
        kfile_edit = rng.randrange(1, 10 ** 6) if rng.random() < 0.35 else None      # seed of the edits of the k-mer file handed to stage 3
        rec = dict(case=i, synth=sy, mutate=mut, stage2=a2, stage3=a3, sharded=sharded, polish=polish, kfile_edit=kfile_edit)
        if only >= 0 and i != only:
            continue
        work = tempfile.mkdtemp(prefix="cf_fuzz_cli_")
        t0 = time.time()
        try:
            report = os.path.join(work, "report.ncrf")
            try:
                _host.synth(report_path=report, pack=False, **sy)
            except Exception as ex:      # (the generator refuses some draws: no read long enough, ...)
                rec.update(identical=True, skipped="generator: " + str(ex)[:120], s=0.0)
                recs.append(rec)
                shutil.rmtree(work, ignore_errors=True)
                continue
            if mut:
                try:
                    fixtures.mutate_report(report, **mut)
                except (ValueError, IndexError) as ex:      # (no stretch shared by that many reads in this report)
                    rec["mutate"] = dict(mut, skipped=str(ex)[:80])
            diffs = []
            outs = {}
            # (this repo's side first, on a short leash: a case with 10^9 pair emissions is minutes on the emulator and hours in the reference — dropped)
            for who, tmpl, extra in (("our", OUR_RUN, dict(root=ROOT, emu=emu)), ("ref", REF_RUN, dict(ref=REF))):
                o2, o3 = os.path.join(work, who, "s2"), os.path.join(work, who, "s3")
                rc, log = run(tmpl % dict(extra, module="distance_based_kmer_recruitment", argv=["--ncrf", report, "--outdir", o2] + a2), 150 if who == "our" else 1800)
                minc = a2[a2.index("--min-coverage") + 1]
                kf, ef = os.path.join(o2, f"unique_kmers_min_edge_cov_{minc}.txt"), os.path.join(o2, f"unique_edges_min_edge_cov_{minc}.txt")
                if rc or not os.path.exists(kf):
                    outs[who] = dict(failed="stage 2", log=log)
                    continue
                kf3 = kf
                if kfile_edit is not None:      # the k-mer list a user hands to stage 3: the same edits of the same lines for both sides
                    kf3 = os.path.join(o2, "edited_kmers.txt")
                    er = random.Random(kfile_edit)
                    kl = open(kf).read().split("\n")
                    er.shuffle(kl)
                    extra_lines = []
                    for x in kl[:50]:
                        what = er.choice(["dup", "lower", "short", "blank", "pad", "N"])
                        extra_lines.append(x if what == "dup" else x.lower() if what == "lower" else x[:-1] if what == "short" else "" if what == "blank" else "  " + x + " \t" if what == "pad" else ("N" + x[1:] if x else x))
                    kl = kl[: max(1, (len(kl) * 4) // 5)] + extra_lines      # (a fifth of the k-mers dropped too)
                    er.shuffle(kl)
                    open(kf3, "w", newline="").write(("\r\n" if er.random() < 0.3 else "\n").join(kl))
                rc, log = run(tmpl % dict(extra, module="read_placer", argv=["--ncrf", report, "--genomic-kmers", kf3, "--outdir", o3] + a3), 150 if who == "our" else 1800)
                pf = os.path.join(o3, "read_positions.csv")
                if rc or not os.path.exists(pf):
                    outs[who] = dict(failed="stage 3", log=log, kmers=open(kf, "rb").read(), edges=sorted(open(ef).read().splitlines()))
                    continue
                lines = open(pf).read().splitlines()
                outs[who] = dict(kmers=open(kf, "rb").read(), edges=sorted(open(ef).read().splitlines()), placed=[x for x in lines if not x.endswith(" None")],
                                 none=sorted(x for x in lines if x.endswith(" None")))
            if sharded and not outs["ref"].get("failed") == "stage 2":
                o2 = os.path.join(work, "sharded", "s2")
                rc, log = run(SHARD_RUN % dict(root=ROOT, sub=str(sharded[1]), world=sharded[0], worker=os.path.join(ROOT, "tests", "sharded_cli_worker.py"),
                                               argv=["--ncrf", report, "--outdir", o2] + a2), 1800)
                minc = a2[a2.index("--min-coverage") + 1]
                kf, ef = os.path.join(o2, f"unique_kmers_min_edge_cov_{minc}.txt"), os.path.join(o2, f"unique_edges_min_edge_cov_{minc}.txt")
                if rc or not os.path.exists(kf) or not os.path.exists(ef):
                    diffs.append("sharded run failed: " + log[-400:])
                else:
                    if open(kf, "rb").read() != outs["ref"]["kmers"]:
                        diffs.append("sharded kmers")
                    if sorted(open(ef).read().splitlines()) != outs["ref"]["edges"]:
                        diffs.append("sharded edges")
            if "placed" in outs["ref"] and polish is not None:
                csv, unit = os.path.join(work, "ref", "s3", "read_positions.csv"), os.path.join(work, "unit.fasta")
                open(unit, "w").write(">u\nACGT\n")
                lo, hi = polish
                pr, po = os.path.join(work, "ref", "pol"), os.path.join(work, "our", "pol")
                rc1, log1 = run(POL_REF % dict(ref=REF, unit=unit, report=report, out=pr, csv=csv, lo=lo, hi="math.inf" if hi is None else str(hi)), 600)
                rc2, log2 = run(POL_OUR % dict(root=ROOT, unit=unit, report=report, out=po, csv=csv, lo=lo, hi_arg=[] if hi is None else ["--max-pos", str(hi)]), 600)
                tr, to = digest_tree(pr), digest_tree(po)
                rec["polisher"] = dict(window=polish, positions=len(tr), ref_rc=rc1, our_rc=rc2)
                if bool(rc1) != bool(rc2) or (not rc1 and tr != to):
                    diffs.append("polisher export: ref rc=%s (%d positions) our rc=%s (%d positions) %s" % (rc1, len(tr), rc2, len(to), (log1[-300:] + " | " + log2[-300:]) if (rc1 or rc2) else ""))
            r, o = outs["ref"], outs["our"]
            if r.get("failed") or o.get("failed"):
                # both may refuse the same input (an empty k-mer set makes the reference's placer fail too); one side only is a difference
                rec["ref_failed"], rec["our_failed"] = r.get("failed"), o.get("failed")
                if r.get("failed") != o.get("failed"):
                    diffs.append("one side failed: ref=%s our=%s" % (r.get("failed"), o.get("failed")))
                rec["logs"] = dict(ref=r.get("log", "")[-600:], our=o.get("log", "")[-600:])
            for what in ("kmers", "edges", "placed", "none"):
                if what in r and what in o and r[what] != o[what]:
                    diffs.append(what)
            rec.update(identical=not diffs, differences=diffs, n_kmers=r.get("kmers", b"").count(b"\n"), n_edges=len(r.get("edges", [])), n_placed=len(r.get("placed", [])),
                       n_none=len(r.get("none", [])))
        except TooSlow as ex:
            rec.update(identical=True, skipped="too large: " + str(ex))
        except Exception as ex:
            rec.update(identical=False, differences=["exception: " + repr(ex)[:300]])
        rec["s"] = round(time.time() - t0, 1)
        if keep and not rec["identical"]:
            shutil.copytree(work, os.path.join(keep, f"case{i}"), dirs_exist_ok=True)
        shutil.rmtree(work, ignore_errors=True)
        recs.append(rec)
        print(json.dumps({k: rec.get(k) for k in ("case", "identical", "differences", "ref_failed", "our_failed", "n_kmers", "n_edges", "n_placed", "n_none", "stage2", "stage3", "mutate", "sharded", "polisher", "kfile_edit", "s")}), flush=True)
    bad = [r for r in recs if not r["identical"]]
    summary = dict(seed=seed, cases=len(recs), identical=len(recs) - len(bad), different=len(bad), both_refused=sum(1 for r in recs if r.get("ref_failed") and r["identical"]), sharded=sum(1 for r in recs if r.get("sharded") and r["identical"] and not r.get("skipped")), polisher_exports=sum(1 for r in recs if r.get("polisher")), edited_kmer_files=sum(1 for r in recs if r.get("kfile_edit") and not r.get("skipped")),
                   placed=sum(r.get("n_placed", 0) for r in recs), edges=sum(r.get("n_edges", 0) for r in recs), seconds=round(time.time() - t_start, 1))
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_cli_vs_reference.json"), str)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


main()
