#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8(f) rank 2: the k-mer OCCURRENCE counts and the most frequent
k-mers that the reference's unit reconstruction starts from
(scripts/better_consensus_unit_reconstruction.py:127-135 get_kmer_counts_reads, :156-167
get_most_frequent_kmers).  Runs the reference itself by import (build container only; edlib and
Biopython, which these two functions never call, are stubbed).

    PYTHONHASHSEED=1 python tests/golden/make_golden_unit_kmers.py          # writes <name>.unit_kmers.json
    PYTHONHASHSEED=2 python tests/golden/make_golden_unit_kmers.py --check
"""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import canon  # noqa: E402
import fixtures  # noqa: E402

KS = (30, 19)


def capture(name, wd):
    sys.dont_write_bytecode = True
    for mod in ("Bio", "Bio.SeqIO", "edlib"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["Bio"].SeqIO = sys.modules["Bio.SeqIO"]
    sys.path.insert(0, "/root/reference/scripts")
    import better_consensus_unit_reconstruction as B
    from ncrf_parser import NCRF_Report
    report = fixtures.make_report(name, wd)
    rep = NCRF_Report(report)
    unit = next(iter(rep.records.values())).motif
    out = dict(fixture=name, report_sha256=fixtures.sha256_file(report), unit_len=len(unit), k={})
    for k in KS:
        counts, top = B.get_most_frequent_kmers(rep, k, unit)
        ranked = sorted(top, key=lambda x: (counts[x], x), reverse=True)
        out["k"][str(k)] = dict(n_distinct=len(counts), total=sum(counts.values()), counts_digest=canon.presence_digest(counts.items()),
                                n_top=len(top), top_digest=canon.set_digest(top), top_head=[[x, counts[x]] for x in ranked[:20]],
                                top_tail=[[x, counts[x]] for x in ranked[-5:]])
    return out


def main():
    check = "--check" in sys.argv
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(fixtures.FIXTURES)
    with tempfile.TemporaryDirectory() as wd:
        for name in names:
            g = capture(name, wd)
            path = os.path.join(HERE, f"{name}.unit_kmers.json")
            if check:
                with open(path) as f:
                    same = json.load(f) == g
                print(name, "IDENTICAL" if same else "DIFFERENT")
                if not same:
                    sys.exit(1)
            else:
                with open(path, "w") as f:
                    json.dump(g, f, indent=0, sort_keys=True)
                print(name, {k: (v["n_distinct"], v["n_top"]) for k, v in g["k"].items()})


if __name__ == "__main__":
    main()
