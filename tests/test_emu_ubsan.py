"""the kernel sources under UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the pool, so the
host emulator is the place: the unmodified .hip files compiled with -fsanitize=undefined (tests/emu/build_emu.sh with
CF_EMU_UBSAN=1) run the distance stage in every table layout, the record sort, A1-A6 and placement of a fixture and the
occurrence counts; any shift past the type width, signed overflow or out-of-range conversion aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "emu", "libcfhip_emu_ubsan.so")

SCRIPT = r'''
import os, sys, tempfile
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import fixtures, pathcheck
from centroflye_amd import _host, _lib
from centroflye_amd.engine import Engine
e = Engine(0, _lib.load(%(lib)r))
assert "libcfhip_emu_ubsan.so" in open("/proc/self/maps").read()
rng = np.random.default_rng(1)
k = rng.integers(0, 2 ** 38, 5000, dtype=np.uint64)
assert np.array_equal(e.selftest_sort(k, 38), np.sort(k))
e.set_param("dist_slots", 2048); e.set_param("dist_block", 128)
pathcheck.check_synthetic_clouds(e)                                                                            # 6-byte slots, multi-chunk postings
pathcheck.check_synthetic_clouds(e, n_reads=1, n_units=330, cloud=3, n_kmers=25, max_d=320, seed=3)           # 16-bit distances
e.set_param("dist_dbits", 6)
pathcheck.check_synthetic_clouds(e, n_reads=3, n_units=60, cloud=4, n_kmers=50, max_d=150, seed=6)            # 6-bit distance field
e.set_param("dist_dbits", 0)
e.set_param("dist_wide", 1)
pathcheck.check_synthetic_clouds(e, n_reads=2, n_units=40, max_d=7, min_d=3, seed=5)                          # 8-byte slots
e.set_param("dist_wide", 0)
pk = _host.parse_report(fixtures.make_report("tiny", tempfile.mkdtemp()))
e.load(pk, 1); e.count_kmers(19); n = e.select_rare(3, 3, 14); e.build_clouds()
ne = e.dist_edges(0, 2 ** 62, 1, 3, 3, 0.8, 0, 1, 400000)
e.sort_edges(); e.edges(ne)
gk = e.kmers()[e.unique_mask().astype(bool)]
e.set_kmers(gk, 19); e.build_clouds(); e.filter_clouds(2)
rank = np.argsort(np.argsort(np.array(pk.ids))).astype(np.int32)
rd, pos, s0, s1 = e.place_reads(pk.classify(50000), rank, 2, 2, 10, 3)
e.count_occurrences(19); e.top_kmers(50)
print("UBSAN-RUN-OK", n, ne, int((pos >= 0).sum()))
'''


def test_kernels_under_ubsan():
    libubsan = subprocess.check_output(["gcc", "-print-file-name=libubsan.so"], text=True).strip()
    if not os.path.isabs(libubsan) or not os.path.exists(libubsan):
        pytest.skip("libubsan not installed")
    subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")], env=dict(os.environ, CF_EMU_UBSAN="1"))
    env = dict(os.environ, LD_PRELOAD=libubsan, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, lib=LIB)], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "UBSAN-RUN-OK" in p.stdout and "runtime error" not in p.stderr, p.stdout[-2000:] + p.stderr[-4000:]
