"""-m gpu: the randomised differential runs of tools/fuzz_parity.py (stage 2 against the OpenMP oracle) and tools/fuzz_place.py (stage 3
against the C placer), tools/fuzz_rr.py (read recruitment against the C restatement and the reference's edlib) and tools/fuzz_unit_kmers.py
(occurrence counts and top-n against the numpy oracle) with fixed seeds and small inputs — two and a half minutes of cases nobody wrote down: other unit lengths, coverages,
error rates, k, rare windows, distances, thresholds, partitions, placer thresholds, and the device knobs that force the rarely taken paths.
Round 5: the first run of the placement one found 500 reads that took 183 s (the contig's overflow map, profiles/
r05_fuzz_place_contig_map_bug.log); longer runs: profiles/r05_fuzz_*.json."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def run_tool(name, args, env_reads, tmp_path, timeout):
    out = tmp_path / "fuzz.json"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", name)] + args + ["--out", str(out)], env=dict(os.environ, CF_FUZZ_READS=env_reads),
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.load(open(out))


def test_stage2_random_cases_equal_the_oracle(tmp_path):
    d = run_tool("fuzz_parity.py", ["200", "--seed", "7", "--seconds", "40"], "300,600,1200", tmp_path, 600)
    s = d["summary"]
    assert s["different"] == 0 and s["identical"] >= 4 and s["with_edges"] >= 2, s
    assert s["identical"] + s["refused"] == s["cases"]


def test_stage3_random_cases_equal_the_c_placer(tmp_path):
    d = run_tool("fuzz_place.py", ["300", "--seed", "7", "--seconds", "30"], "200,500,1000", tmp_path, 600)
    s = d["summary"]
    assert s["different"] == 0 and s["identical"] >= 10 and s["placed"] >= 1000, s
    # no case may take the device longer than a few milliseconds per read (the contig-map bug: 80 - 900 ms per read)
    slow = [(c["case"], c["synth"]["n_reads"], c["place_ms"]) for c in d["cases"] if c.get("place_ms") and c["knobs"].get("place_grid", 128) >= 16
            and c["place_ms"] > 20.0 * c["synth"]["n_reads"]]
    assert not slow, slow


def test_read_recruitment_random_batches_equal_the_restatement_and_edlib(tmp_path):
    d = run_tool("fuzz_rr.py", ["100000", "--seed", "7", "--seconds", "10"], "", tmp_path, 300)
    s = d["summary"]
    assert s["different"] == 0 and s["identical"] == s["cases"] >= 50 and s["distances_within_threshold"] > 100, s


def test_occurrence_counts_and_top_n_random_cases_equal_the_oracle(tmp_path):
    d = run_tool("fuzz_unit_kmers.py", ["1000", "--seed", "7", "--seconds", "15"], "20,100,400", tmp_path, 300)
    s = d["summary"]
    assert s["different"] == 0 and s["identical"] >= 2, s
