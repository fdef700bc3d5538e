import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fx_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("fixtures"))


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with open(os.path.join(ROOT, "tests", "golden", f"{name}.json")) as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def report(fx_dir):
    import fixtures
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = fixtures.make_report(name, fx_dir)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def oracle_stage2(report):
    """numpy-oracle stage 2 of a fixture (cached per session; 10-20 s each)."""
    import fixtures
    from oracle import ncrf, recruit
    cache = {}

    def get(name, **override):
        key = (name, tuple(sorted(override.items())))
        if key not in cache:
            p2 = fixtures.stage2_params(name)
            p2.update(override)
            records, alns, lens = ncrf.parse_report(report(name))
            res = recruit.stage2(records, k=p2["k"], bottom=p2["bottom"], top=p2["top"], coverage=p2["coverage"],
                                 kmer_survival_rate=p2["kmer_survival_rate"], max_nonuniq=p2["max_nonuniq"],
                                 min_n=p2["min_nreads"], max_n=p2["max_nreads"], min_d=p2["min_distance"],
                                 max_d=p2["max_distance"], min_coverage=p2["min_coverage"])
            cache[key] = (records, alns, lens, res, p2)
        return cache[key]
    return get


@pytest.fixture(scope="session")
def emu_lib():
    """The kernel sources compiled against the host emulator (tests/emu) — logic checks on CPU."""
    from centroflye_amd import _lib
    script = os.path.join(ROOT, "tests", "emu", "build_emu.sh")
    subprocess.check_call(["bash", script])
    return _lib.load(os.path.join(ROOT, "tests", "emu", "libcfhip_emu.so"))


def lines_from_placement(ids, rd, pos, s0, s1):
    out = []
    for a, b, c, d in zip(rd, pos, s0, s1):
        if c < 0 and b == 0:
            out.append(f"{ids[a]} 0")
        elif b < 0:
            out.append(f"{ids[a]} None")
        else:
            out.append(f"{ids[a]} {b} {c} {d}")
    return out
