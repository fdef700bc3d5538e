"""Stage 3 with --n-motif 2 against the REFERENCE's own outputs (tests/golden/<fixture>.n_motif2.json, written by
tests/golden/make_golden_nmotif2.py importing /root/reference/scripts): units are pairs of motif copies (ncrf_parser.py:28-59 with
motif * n; read_placer.py:141-142 passes the flag through).  Round 4 had this path only against the oracle, and the oracle's n = 2 unit
split pinned on five records per fixture; here every record's unit columns, the clouds before and after the multiplicity filter and
every line of read_positions.csv are the reference's — oracle and host packer on CPU, the stage script on the emulated kernels, and
(-m gpu) `python scripts/read_placer.py --n-motif 2` on the MI355X."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import canon
import fixtures
from centroflye_amd import _host, read_placer, session
from centroflye_amd.engine import Engine
from oracle import ncrf, placer, recruit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(fixtures.FIXTURES)


def g2(name):
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.n_motif2.json")) as f:
        return json.load(f)


def _argv(report, kfile, outdir, p3):
    return ["--ncrf", report, "--genomic-kmers", kfile, "--outdir", outdir, "--n-motif", "2", "--min-cloud-kmer-freq", str(p3["min_cloud_kmer_freq"]),
            "--min-kmer-mult", str(p3["min_kmer_mult"]), "--min-unit", str(p3["min_unit"]), "--min-inters", str(p3["min_inters"]),
            "--prefix-threshold", str(p3["prefix_threshold"])]


def _check_lines(path, g):
    with open(path) as f:
        lines = f.read().splitlines()
    assert [ln for ln in lines if not ln.endswith(" None")] == g["read_positions"]["placed"]      # byte for byte, in order
    assert sorted(ln for ln in lines if ln.endswith(" None")) == g["read_positions"]["none"]      # (the reference's order there is hash-seed dependent)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_and_packer_against_the_reference(name, report):
    g = g2(name)
    assert fixtures.sha256_file(report(name)) == g["report_sha256"]
    records, alns, lens = ncrf.parse_report(report(name))
    pk = _host.parse_report(report(name))
    up2, _, _, uc2 = pk.units(2)
    assert list(records) == list(g["unit_cols_n2"]) == pk.ids
    for r, (r_id, rec) in enumerate(records.items()):
        assert ncrf.unit_columns(rec, 2) == g["unit_cols_n2"][r_id]
        c2 = uc2[up2[r]:up2[r + 1]]
        assert [int(c[0]) for c in c2] + ([int(c2[-1][1])] if len(c2) else []) == g["unit_cols_n2"][r_id]
    p3 = g["stage3"]
    kfile = os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt")
    assert fixtures.sha256_file(kfile) == g["unique_kmers_sha256"]
    with open(kfile) as f:
        gk = np.array(sorted(recruit.encode_kmer(x.strip()) for x in f if x.strip()), dtype=np.uint64)
    r3 = placer.stage3(records, alns, lens, gk, n_motif=2, k_cloud=p3["k_cloud"], min_cloud_kmer_freq=p3["min_cloud_kmer_freq"], min_kmer_mult=p3["min_kmer_mult"],
                       min_unit=p3["min_unit"], min_inters=p3["min_inters"], prefix_threshold=p3["prefix_threshold"])
    ks = [recruit.decode_kmer(c, p3["k_cloud"]) for c in gk]
    up = r3["unit_ptr"]

    def as_clouds(cp, ent):
        return [[[ks[i] for i in ent[cp[u]:cp[u + 1]]] for u in range(up[r], up[r + 1])] for r in range(len(up) - 1)]
    c3, c3f = as_clouds(r3["cloud_ptr"], r3["entries"]), as_clouds(r3["f_cloud_ptr"], r3["f_entries"])
    assert [[len(c) for c in units] for units in c3] == g["clouds3"]["sizes"] and canon.clouds_digest(c3) == g["clouds3"]["digest"]
    assert [[len(c) for c in units] for units in c3f] == g["clouds3_filtered"]["sizes"] and canon.clouds_digest(c3f) == g["clouds3_filtered"]["digest"]
    assert [ln for ln in r3["lines"] if not ln.endswith(" None")] == g["read_positions"]["placed"]
    assert sorted(ln for ln in r3["lines"] if ln.endswith(" None")) == g["read_positions"]["none"]


@pytest.mark.parametrize("name", ["lowcov", "hor2055"])
def test_stage_script_on_emulated_kernels(name, emu_lib, report, tmp_path):
    g = g2(name)
    session.reset()
    session._engine = Engine(0, emu_lib)
    try:
        read_placer.main(_argv(report(name), os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt"), str(tmp_path), g["stage3"]))
        _check_lines(tmp_path / "read_positions.csv", g)
    finally:
        session.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_stage_script_on_the_gpu(name, report, tmp_path):
    """`python -u scripts/read_placer.py ... --n-motif 2` as centroFlye.py:196-204 would spawn it with that flag."""
    g = g2(name)
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "read_placer.py")] +
                          _argv(report(name), os.path.join(ROOT, "tests", "golden", f"{name}.unique_kmers.txt"), str(tmp_path), g["stage3"]), stdout=subprocess.DEVNULL)
    _check_lines(tmp_path / "read_positions.csv", g)
    assert len(g["read_positions"]["placed"]) >= 2
