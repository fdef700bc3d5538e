"""SURVEY.md §8(f) rank 3 — per-position read-unit export (reference scripts/eltr_polisher.py:53-97): the oracle and the
compiled exporter (libcfhost.so, through the mirror module and through the CLI with the reference's script name)
against golden vectors captured by running the reference itself (tests/golden/make_golden_polisher.py).  CPU only."""
import hashlib
import json
import math
import os
import subprocess
import sys
import types

import pytest

import fixtures
from oracle import ncrf, polisher

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = list(fixtures.FIXTURES)


def _golden(name):
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.read_units.json")) as f:
        return json.load(f)


def _csv(name, tmp_path):
    with open(os.path.join(ROOT, "tests", "golden", f"{name}.json")) as f:
        g = json.load(f)
    path = os.path.join(tmp_path, f"{name}.read_positions.csv")
    with open(path, "w") as f:
        f.write("".join(ln + "\n" for ln in g["read_positions"]["placed"] + g["read_positions"]["none"]))
    return path


def _digest_tree(outdir):
    res = {}
    for d in sorted(os.listdir(outdir)):
        if d.startswith("pos_"):
            ent = []
            for fn in ("read_units.fasta", "median_read_unit.fasta"):
                with open(os.path.join(outdir, d, fn), "rb") as f:
                    data = f.read()
                ent.append([hashlib.sha256(data).hexdigest(), len(data)])
            res[d[4:]] = ent
    return res


@pytest.mark.parametrize("name", NAMES)
def test_oracle_against_reference_golden(name, report, tmp_path):
    g = _golden(name)
    assert fixtures.sha256_file(report(name)) == g["report_sha256"]
    records, _, _ = ncrf.parse_report(report(name))
    placement = polisher.read_reported_positions(_csv(name, str(tmp_path)))
    for w in g["windows"]:
        files = polisher.export(records, placement, w["min_pos"], math.inf if w["max_pos"] is None else w["max_pos"])
        got = {str(p): [[hashlib.sha256(t.encode()).hexdigest(), len(t)] for t in texts] for p, texts in files.items()}
        assert got == w["files"]
        assert len(files) == w["n_positions"]
        if w["first_position"] is not None:
            assert files[w["first_position"]][0][:400] == w["first_file_head"]


@pytest.mark.parametrize("name", NAMES)
def test_native_export_against_reference_golden(name, report, tmp_path):
    from centroflye_amd import eltr_polisher
    g = _golden(name)
    unit = os.path.join(str(tmp_path), "unit.fasta")
    open(unit, "w").write(">u\nACGT\n")
    csv = _csv(name, str(tmp_path))
    for i, w in enumerate(g["windows"]):
        outdir = os.path.join(str(tmp_path), f"polish{i}")
        params = types.SimpleNamespace(unit=unit, ncrf=report(name), outdir=outdir, read_placement=csv, min_pos=w["min_pos"],
                                       max_pos=math.inf if w["max_pos"] is None else w["max_pos"])
        pol = eltr_polisher.ELTR_Polisher(params)
        assert pol.max_pos == w["resolved_max_pos"]
        pos2read = pol.map_pos2read()
        assert (len(pos2read), sum(len(v) for v in pos2read.values())) == (w["n_positions"], w["n_units"])
        files = pol.export_read_units(pos2read)
        assert sorted(files) == sorted(pos2read)
        assert _digest_tree(outdir) == w["files"]


def test_cli_with_the_reference_script_name(report, tmp_path):
    g = _golden("lowcov")
    unit = os.path.join(str(tmp_path), "unit.fasta")
    open(unit, "w").write(">u\nACGT\n")
    outdir = os.path.join(str(tmp_path), "polishing")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "eltr_polisher.py"), "--read-placement", _csv("lowcov", str(tmp_path)),
           "--unit", unit, "--ncrf", report("lowcov"), "--outdir", outdir, "--export-only"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _digest_tree(outdir) == g["windows"][0]["files"]


def test_missing_unit_file_is_an_error(report, tmp_path):
    from centroflye_amd import eltr_polisher
    params = types.SimpleNamespace(unit=os.path.join(str(tmp_path), "nope.fasta"), ncrf=report("tiny"), outdir=str(tmp_path),
                                   read_placement=_csv("tiny", str(tmp_path)), min_pos=0, max_pos=math.inf)
    with pytest.raises(FileNotFoundError):
        eltr_polisher.ELTR_Polisher(params)
