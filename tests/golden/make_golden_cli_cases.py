#!/usr/bin/env python3
"""Capture what the REFERENCE's two command lines write for random small reports with random options (build container only: needs
/root/reference).  Input: the record of a tools/fuzz_cli_vs_reference.py run (its cases carry the generator parameters, the mutations and
the options of both stages); the picked cases are run again through /root/reference/scripts/distance_based_kmer_recruitment.py and
read_placer.py (Biopython stubbed) and tests/golden/cli_cases.json gets, per case: the report's SHA-256, the k-mer file's
SHA-256, the digest of the sorted edge lines, the placed lines in order and the None lines sorted.  tests/test_cli_cases.py runs this
repo's command lines on the same reports — a few cases on the emulated kernels in the CPU suite, all of them on the GPU.

    python tests/golden/make_golden_cli_cases.py --from /tmp/fzcli_seed3.json [--pick 3,8,21] [--n 12]
"""
import argparse, hashlib, json, os, random, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import canon  # noqa: E402
import fixtures  # noqa: E402
from centroflye_amd import _host  # noqa: E402

REF = "/root/reference/scripts"
REF_RUN = r'''
import sys, types
sys.dont_write_bytecode = True
bio = types.ModuleType("Bio"); bio.SeqIO = types.ModuleType("Bio.SeqIO"); sys.modules["Bio"] = bio; sys.modules["Bio.SeqIO"] = bio.SeqIO
sys.path.insert(0, %(ref)r)
import %(module)s as M
sys.argv = [%(module)r] + %(argv)r
M.main()
'''


def write_report(case, path):
    """The case's report: the seeded native generator, then the seeded mutations (soft-masked stretches, N calls)."""
    _host.synth(report_path=path, pack=False, **case["synth"])
    m = case.get("mutate")
    if m and "skipped" not in m:
        fixtures.mutate_report(path, **{k: v for k, v in m.items() if k != "skipped"})
    return path


def reference_outputs(case, report, work):
    o2, o3 = os.path.join(work, "s2"), os.path.join(work, "s3")
    env = dict(os.environ, PYTHONHASHSEED=str(random.randrange(1, 10 ** 6)))
    for module, argv in (("distance_based_kmer_recruitment", ["--ncrf", report, "--outdir", o2] + case["stage2"]),):
        subprocess.run([sys.executable, "-c", REF_RUN % dict(ref=REF, module=module, argv=argv)], check=True, capture_output=True, env=env)
    minc = case["stage2"][case["stage2"].index("--min-coverage") + 1]
    kf, ef = os.path.join(o2, f"unique_kmers_min_edge_cov_{minc}.txt"), os.path.join(o2, f"unique_edges_min_edge_cov_{minc}.txt")
    subprocess.run([sys.executable, "-c", REF_RUN % dict(ref=REF, module="read_placer", argv=["--ncrf", report, "--genomic-kmers", kf, "--outdir", o3] + case["stage3"])],
                   check=True, capture_output=True, env=env)
    kdata = open(kf, "rb").read()
    elines = open(ef).read().splitlines()
    lines = open(os.path.join(o3, "read_positions.csv")).read().splitlines()
    return dict(unique_kmers=dict(sha256=hashlib.sha256(kdata).hexdigest(), n=kdata.count(b"\n"), with_N=sum(1 for ln in kdata.split(b"\n") if b"N" in ln)),
                edges=dict(n=len(elines), digest=canon.edge_lines_digest(elines)),
                read_positions=dict(placed=[x for x in lines if not x.endswith(" None")], none=sorted(x for x in lines if x.endswith(" None"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="src", required=True); ap.add_argument("--pick", default=""); ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--check", action="store_true", help="compare with the committed file instead of writing it")
    a = ap.parse_args()
    cases = [c for c in json.load(open(a.src))["cases"] if c.get("identical") and not c.get("skipped") and not c.get("ref_failed")]
    if a.pick:
        want = [int(x) for x in a.pick.split(",")]
        picked = [c for c in cases if c["case"] in want]
    else:      # the quick ones with something in every file first, then variety: mutated reports, --n-motif 2, read windows, other rare windows
        def score(c):
            s2, s3 = c["stage2"], c["stage3"]
            return (bool(c["n_edges"]) + bool(c["n_placed"]) + bool(c["n_none"]) + bool(c.get("mutate")) + ("--min-nreads" in s2) + ("--bottom" in s2)
                    + (s3[s3.index("--n-motif") + 1] == "2")) - c["s"] / 30.0
        picked = sorted(cases, key=score, reverse=True)[:a.n]
    out = []
    for c in sorted(picked, key=lambda c: c["case"]):
        with tempfile.TemporaryDirectory() as work:
            report = write_report(c, os.path.join(work, "report.ncrf"))
            g = dict(case=c["case"], synth=c["synth"], mutate=c.get("mutate"), stage2=c["stage2"], stage3=c["stage3"], report_sha256=fixtures.sha256_file(report),
                     seconds_in_the_fuzz_run=c["s"], **reference_outputs(c, report, work))
        out.append(g)
        print(c["case"], g["unique_kmers"], g["edges"]["n"], len(g["read_positions"]["placed"]), len(g["read_positions"]["none"]), flush=True)
    path = os.path.join(HERE, "cli_cases.json")
    doc = dict(what="outputs of the reference's own scripts (tests/golden/make_golden_cli_cases.py) on random small reports with random options", cases=out)
    if a.check:
        old = json.load(open(path))
        same = [o for o in old["cases"] for n in out if n["case"] == o["case"] and {k: v for k, v in n.items() if k != "seconds_in_the_fuzz_run"} == {k: v for k, v in o.items() if k != "seconds_in_the_fuzz_run"}]
        print("IDENTICAL" if len(same) == len(out) else "DIFFERENT")
        sys.exit(0 if len(same) == len(out) else 1)
    json.dump(doc, open(path, "w"), indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


main()
