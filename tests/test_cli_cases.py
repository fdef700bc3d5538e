"""Both drop-in command lines on RANDOM small reports with RANDOM options against what the reference's own scripts wrote for them
(tests/golden/cli_cases.json, captured by tests/golden/make_golden_cli_cases.py from a tools/fuzz_cli_vs_reference.py run): split records,
both strands, soft-masked stretches and N calls in some reports; other k, rare windows (--bottom / --top / --kmer-survival-rate),
--max-nonuniq, --min-coverage, distances, read windows (--min-nreads / --max-nreads), --n-motif 2, placer thresholds.  The five fixtures
of tests/golden/*.json pin the defaults at depth; these pin the options' meaning.  CPU: the quickest cases on the emulated kernels;
-m gpu: every case through scripts/*.py as centroFlye.py spawns them."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

import canon
import fixtures
from centroflye_amd import _host, distance_based_kmer_recruitment as dbkr, read_placer, session
from centroflye_amd.engine import Engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "cli_cases.json")) as _f:
    CASES = json.load(_f)["cases"]
QUICK = sorted(CASES, key=lambda c: c["seconds_in_the_fuzz_run"])[:9]


def _report(case, d):
    path = os.path.join(str(d), f"case{case['case']}.ncrf")
    _host.synth(report_path=path, pack=False, **case["synth"])
    m = case.get("mutate")
    if m and "skipped" not in m:
        fixtures.mutate_report(path, **m)
    assert fixtures.sha256_file(path) == case["report_sha256"], "the generator (or the mutation) no longer writes the report the golden was taken on"
    return path


def _check(case, out2, out3):
    minc = case["stage2"][case["stage2"].index("--min-coverage") + 1]
    with open(os.path.join(out2, f"unique_kmers_min_edge_cov_{minc}.txt"), "rb") as f:
        assert hashlib.sha256(f.read()).hexdigest() == case["unique_kmers"]["sha256"]
    with open(os.path.join(out2, f"unique_edges_min_edge_cov_{minc}.txt")) as f:
        elines = f.read().splitlines()
    assert len(elines) == case["edges"]["n"] and canon.edge_lines_digest(elines) == case["edges"]["digest"]
    with open(os.path.join(out3, "read_positions.csv")) as f:
        lines = f.read().splitlines()
    assert [x for x in lines if not x.endswith(" None")] == case["read_positions"]["placed"]
    assert sorted(x for x in lines if x.endswith(" None")) == case["read_positions"]["none"]


def test_the_cases_cover_the_options():
    s2 = [c["stage2"] for c in CASES]
    s3 = [c["stage3"] for c in CASES]
    assert len(CASES) >= 10
    assert any("--bottom" in a for a in s2) and any("--min-nreads" in a for a in s2) and any(a[a.index("--n-motif") + 1] == "2" for a in s3)
    assert any(c.get("mutate") for c in CASES) and any(c["edges"]["n"] for c in CASES)
    assert any(c["read_positions"]["placed"] for c in CASES) and any(c["read_positions"]["none"] for c in CASES)
    assert len({a[a.index("-k") + 1] for a in s2}) >= 3


@pytest.mark.parametrize("case", QUICK, ids=lambda c: f"case{c['case']}")
def test_cli_case_on_emulated_kernels(case, emu_lib, tmp_path):
    session.reset()
    session._engine = Engine(0, emu_lib)
    session._engine.set_param("dist_slots", 2048); session._engine.set_param("dist_block", 128)
    try:
        report = _report(case, tmp_path)
        out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
        dbkr.main(["--ncrf", report, "--outdir", out2] + case["stage2"])
        minc = case["stage2"][case["stage2"].index("--min-coverage") + 1]
        read_placer.main(["--ncrf", report, "--genomic-kmers", os.path.join(out2, f"unique_kmers_min_edge_cov_{minc}.txt"), "--outdir", out3] + case["stage3"])
        _check(case, out2, out3)
    finally:
        session.reset()


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"case{c['case']}")
def test_cli_case_on_the_gpu(case, tmp_path):
    report = _report(case, tmp_path)
    out2, out3 = str(tmp_path / "s2"), str(tmp_path / "s3")
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "distance_based_kmer_recruitment.py"), "--ncrf", report, "--outdir", out2] + case["stage2"],
                          stdout=subprocess.DEVNULL)
    minc = case["stage2"][case["stage2"].index("--min-coverage") + 1]
    subprocess.check_call([sys.executable, "-u", os.path.join(ROOT, "scripts", "read_placer.py"), "--ncrf", report, "--genomic-kmers",
                           os.path.join(out2, f"unique_kmers_min_edge_cov_{minc}.txt"), "--outdir", out3] + case["stage3"], stdout=subprocess.DEVNULL)
    _check(case, out2, out3)
