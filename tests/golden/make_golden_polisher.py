#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8(f) rank 3: the per-position read-unit FASTA files the reference's polisher
writes before it calls Flye (scripts/eltr_polisher.py:53-66 map_pos2read, :68-97 export_read_units).  Runs the
reference itself by import (build container only; edlib and Biopython, which these methods never call, are stubbed,
and read_bio_seq — the unit FASTA is not used by the export — is replaced by a constant).

    PYTHONHASHSEED=1 python tests/golden/make_golden_polisher.py          # writes <name>.read_units.json
    PYTHONHASHSEED=2 python tests/golden/make_golden_polisher.py --check
"""
import hashlib
import json
import math
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402

WINDOWS = ((0, math.inf), (2, 6))     # default run and an explicit --min-pos / --max-pos window


def placement_lines(name):
    with open(os.path.join(HERE, f"{name}.json")) as f:
        g = json.load(f)
    return g["read_positions"]["placed"] + g["read_positions"]["none"]


def digest_tree(outdir):
    res = {}
    for d in sorted(os.listdir(outdir)):
        if not d.startswith("pos_"):
            continue
        ent = []
        for fn in ("read_units.fasta", "median_read_unit.fasta"):
            with open(os.path.join(outdir, d, fn), "rb") as f:
                data = f.read()
            ent.append([hashlib.sha256(data).hexdigest(), len(data)])
        res[d[4:]] = ent
    return res


def capture(name, wd):
    sys.dont_write_bytecode = True
    for mod in ("Bio", "Bio.SeqIO", "edlib"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.path.insert(0, "/root/reference/scripts")
    import eltr_polisher as E
    E.read_bio_seq = lambda fn: "ACGT"
    report = fixtures.make_report(name, wd)
    csv = os.path.join(wd, f"{name}.read_positions.csv")
    with open(csv, "w") as f:
        f.write("".join(ln + "\n" for ln in placement_lines(name)))
    unit = os.path.join(wd, "unit.fasta")
    open(unit, "w").write(">u\nACGT\n")
    out = dict(fixture=name, report_sha256=fixtures.sha256_file(report), windows=[])
    for w, (lo, hi) in enumerate(WINDOWS):
        outdir = os.path.join(wd, f"{name}.polish{w}")
        params = types.SimpleNamespace(unit=unit, ncrf=report, outdir=outdir, read_placement=csv, min_pos=lo, max_pos=hi)
        pol = E.ELTR_Polisher(params)
        pos2read = pol.map_pos2read()
        files = pol.export_read_units(pos2read)
        first = min(pos2read) if pos2read else None
        sample = open(files[first][0]).read()[:400] if first is not None else ""
        out["windows"].append(dict(min_pos=lo, max_pos=None if hi == math.inf else hi, resolved_max_pos=pol.max_pos,
                                   n_positions=len(pos2read), n_units=sum(len(v) for v in pos2read.values()),
                                   files=digest_tree(outdir), first_position=first, first_file_head=sample))
    return out


def main():
    check = "--check" in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(fixtures.FIXTURES)
    with tempfile.TemporaryDirectory() as wd:
        for name in names:
            g = capture(name, wd)
            path = os.path.join(HERE, f"{name}.read_units.json")
            if check:
                with open(path) as f:
                    same = json.load(f) == g
                print(name, "IDENTICAL" if same else "DIFFERENT")
                if not same:
                    sys.exit(1)
            else:
                with open(path, "w") as f:
                    json.dump(g, f, indent=1)
                print(name, [(w["n_positions"], w["n_units"], w["resolved_max_pos"]) for w in g["windows"]])


if __name__ == "__main__":
    main()
