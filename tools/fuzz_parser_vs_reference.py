#!/usr/bin/env python3
"""Developer tool (BUILD CONTAINER ONLY: needs /root/reference, no GPU): randomised differential test of row A0 — the compiled NCRF-report
ingestion (centroflye_amd/csrc/host/cfhost.cpp: parse, keep the longest alignment >= 5000 per read, reverse-complement '-' records, unit
split for n = 1 and 2, classify) against the reference's ncrf_parser.NCRF_Report (regexes, ncrf_parser.py:28-145).  Every case writes a
random small report with the native generator (unit length, units per read, split records so that reads have several alignments, short
records around the 5 000 limit) and then edits its TEXT in ways NCRF output varies: runs of blanks and tabs between fields, comment and
empty lines, lower-case stretches in the read row AND in the motif row, N calls, Windows line ends; the record order, the per-record
fields, both oriented rows, the unit columns for n = 1 and n = 2, the classes under a random --prefix-threshold and the discarded reads
must be equal.  usage: tools/fuzz_parser_vs_reference.py [cases] [--seed S] [--seconds T] [--keep DIR]"""
import json, os, random, shutil, sys, tempfile, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/scripts"


def arg(name, default, conv=int):
    return conv(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def edit_text(path, rng):
    with open(path) as f:
        lines = f.read().split("\n")
    out, what = [], []
    eol = "\r\n" if rng.random() < 0.15 else "\n"
    for ln in lines:
        if ln and not ln.startswith("#"):
            head = ln.split(None, 4)
            row = head[4] if len(head) == 5 else ""
            if rng.random() < 0.3 and row:      # lower-case stretch (read row or motif row alike), or an N call
                row = list(row)
                c, n = rng.randrange(len(row)), rng.choice([1, 3, 40])
                kind = rng.choice(["lower", "lower", "N"])
                for j in range(c, min(len(row), c + n)):
                    if row[j] != "-":
                        row[j] = row[j].lower() if kind == "lower" else ("N" if head[0][-1] not in "+-" else row[j].lower())
                row = "".join(row)
                what.append(kind)
            seps = [rng.choice([" ", " ", "  ", "\t", "   "]) for _ in range(4)]
            ln = "".join(h + s for h, s in zip(head[:4], seps)) + row
            if rng.random() < 0.1:
                ln = ln + rng.choice([" ", "  "])      # trailing blanks (the reference strips the line)
        out.append(ln)
        if rng.random() < 0.05:
            out.append(rng.choice(["", "# a comment", "   ", "#"]))
    with open(path, "w", newline="") as f:
        f.write(eol.join(out))
    return sorted(set(what)) + (["crlf"] if eol != "\n" else [])


def main():
    if not os.path.isdir(REF):
        sys.exit("this tool needs /root/reference (build container only)")
    sys.dont_write_bytecode = True
    bio = types.ModuleType("Bio"); bio.SeqIO = types.ModuleType("Bio.SeqIO"); sys.modules["Bio"] = bio; sys.modules["Bio.SeqIO"] = bio.SeqIO
    sys.path.insert(0, REF)
    import ncrf_parser as RP
    import numpy as np
    from centroflye_amd import _host
    from centroflye_amd.ncrf_parser import NCRF_Report as ONR
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
    seed, budget, keep = arg("--seed", 1), arg("--seconds", 10 ** 9, float), arg("--keep", "", str)
    rng = random.Random(seed)
    recs, t_start = [], time.time()
    for i in range(n_cases):
        if time.time() - t_start > budget:
            break
        unit_len = rng.choice([60, 120, 200, 342, 684, 2055])
        upr = rng.choice([3, 8, 20, 40])
        sy = dict(seed=rng.randrange(1, 1 << 30), unit_len=unit_len, monomer_len=rng.choice([unit_len // 4, unit_len // 2, unit_len // 3]), n_units=rng.choice([20, 60]),
                  flank=rng.choice([10000, 60000]), n_reads=rng.choice([10, 25, 60]), mean_len=float(max(5200, unit_len * upr)), sigma=rng.choice([0.1, 0.4]),
                  min_len=rng.choice([4000, 5100]), max_len=int(max(6000, 3 * unit_len * upr)), min_aligned=rng.choice([3000, 4800, 5000]), n_prefix=rng.choice([0, 2]),
                  n_suffix=rng.choice([0, 3]), prefix_threshold=rng.choice([5000, 50000]), p_split=rng.choice([0.0, 0.2, 0.5]), p_sub=rng.choice([0.005, 0.03]),
                  p_del=rng.choice([0.005, 0.03]), p_ins=rng.choice([0.005, 0.02]), var_len=rng.choice([1, 8]))
        thr = rng.choice([1000, 5000, 50000, 200000])
        rec = dict(case=i, synth=sy, prefix_threshold=thr)
        work = tempfile.mkdtemp(prefix="cf_fuzz_parse_")
        try:
            report = os.path.join(work, "report.ncrf")
            try:
                _host.synth(report_path=report, pack=False, **sy)
            except _host.HostError as ex:
                rec.update(identical=True, skipped="generator: " + str(ex)[:100])
                recs.append(rec)
                continue
            rec["edits"] = edit_text(report, rng)
            diffs = []
            ref = RP.NCRF_Report(report)
            pk = _host.parse_report(report, keep_rows=True)
            if pk.ids != list(ref.records):
                diffs.append("record order / kept reads")
            else:
                for n in (1, 2):
                    up, us, ue, uc = pk.units(n)
                    for r, (r_id, x) in enumerate(ref.records.items()):
                        if n == 1:
                            m = pk.meta[r]
                            if (int(m[0]), int(m[1]), int(m[2]), int(m[3]), "+-"[int(m[4])]) != (x.r_len, x.r_al_len, x.r_st, x.r_en, x.strand):
                                diffs.append(f"fields of {r_id}")
                            if pk.row(r, 0) != x.r_al or pk.row(r, 1) != x.m_al:
                                diffs.append(f"rows of {r_id}")
                            if pk.bases[pk.read_off[r]:pk.read_off[r + 1]].tobytes().decode() != x.r_al.replace("-", ""):
                                diffs.append(f"bases of {r_id}")
                        mas = x.get_motif_alignments(n=n)
                        want = [ma.start for ma in mas] + ([mas[-1].end] if mas else [])
                        cols = uc[up[r]:up[r + 1]]
                        got = [int(c[0]) for c in cols] + ([int(cols[-1][1])] if len(cols) else [])
                        if got != want:
                            diffs.append(f"unit columns n={n} of {r_id}")
                pre, mid, suf = ref.classify(large_threshold=thr)
                cls = pk.classify(thr)
                for c, want in ((0, pre), (1, mid), (2, suf)):
                    if [pk.ids[j] for j in np.flatnonzero(cls == c)] != want:
                        diffs.append(f"class {c}")
            if sorted(pk.discarded_reads) != sorted(ref.discarded_reads):
                diffs.append("discarded reads")
            # the same through the drop-in class (centroflye_amd.ncrf_parser.NCRF_Report): records, their attributes, the MotifAlignment tuples
            mir = ONR(report)
            if list(mir.records) != list(ref.records) or sorted(mir.discarded_reads) != sorted(ref.discarded_reads):
                diffs.append("mirror: records / discarded")
            else:
                for r_id, x in ref.records.items():
                    y = mir.records[r_id]
                    if (y.r_id, y.r_len, y.r_al_len, y.r_st, y.r_en, y.strand, y.motif, y.r_al, y.m_al) != (x.r_id, x.r_len, x.r_al_len, x.r_st, x.r_en, x.strand, x.motif, x.r_al, x.m_al):
                        diffs.append(f"mirror: attributes of {r_id}")
                    for n in (1, 2):
                        if [tuple(m) for m in y.get_motif_alignments(n=n)] != [tuple(m) for m in x.get_motif_alignments(n=n)]:
                            diffs.append(f"mirror: motif alignments n={n} of {r_id}")
                if tuple(mir.classify(thr)) != tuple(ref.classify(large_threshold=thr)):
                    diffs.append("mirror: classify")
                if any(mir.read_lens[r] != ref.read_lens[r] for r in mir.read_lens):
                    diffs.append("mirror: read_lens")
            rec.update(identical=not diffs, differences=diffs[:6], n_records=len(ref.records), n_discarded=len(ref.discarded_reads),
                       n_units=int(pk.units(1)[0][-1]) if pk.n_reads else 0)
        except Exception as ex:
            rec.update(identical=False, differences=["exception: " + repr(ex)[:300]])
        if keep and not rec["identical"]:
            shutil.copytree(work, os.path.join(keep, f"case{i}"), dirs_exist_ok=True)
        shutil.rmtree(work, ignore_errors=True)
        recs.append(rec)
        if not rec["identical"]:
            print("DIFFERENCE:", json.dumps(rec), flush=True)
    bad = [r for r in recs if not r["identical"]]
    summary = dict(seed=seed, cases=len(recs), identical=len(recs) - len(bad), different=len(bad), skipped=sum(1 for r in recs if r.get("skipped")),
                   records=sum(r.get("n_records", 0) for r in recs), discarded=sum(r.get("n_discarded", 0) for r in recs), units=sum(r.get("n_units", 0) for r in recs),
                   seconds=round(time.time() - t_start, 1))
    out = arg("--out", os.path.join(ROOT, "gpurun_out", "fuzz_parser_vs_reference.json"), str)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(dict(summary=summary, cases=recs), open(out, "w"), indent=1)
    print(json.dumps(summary))
    sys.exit(1 if bad else 0)


main()
