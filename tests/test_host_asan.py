"""the host C++ (libcfhost.so: NCRF parser, unit split, classifier, generator, file writers, read-unit
exporter) under AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C centroflye_amd/csrc asan` builds the library;
a child interpreter with libasan preloaded runs the host tests on it, then parses a few hundred corrupted reports
(truncations, byte flips, deleted lines): every one must end in a result or a HostError, never in a sanitizer report.
CPU only (GPU AddressSanitizer is not available on the pool)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASAN_LIB = os.path.join(ROOT, "centroflye_amd", "csrc", "build", "libcfhost_asan.so")

FUZZ = r'''
import os, sys, random
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from centroflye_amd import _host
assert "asan" in _host._LIB_PATH
maps = open("/proc/self/maps").read()
_host.lib()
assert "libcfhost_asan.so" in open("/proc/self/maps").read()
import fixtures
d = %(tmp)r
src = fixtures.make_report("lowcov", d)
data = open(src, "rb").read()
rng = random.Random(7)
ok = bad = 0
for i in range(300):
    b = bytearray(data[: rng.randrange(2000, 60000)])
    for _ in range(rng.randrange(0, 6)):
        what = rng.randrange(4)
        pos = rng.randrange(len(b))
        if what == 0: b[pos] = rng.randrange(256)
        elif what == 1: del b[pos: pos + rng.randrange(1, 200)]
        elif what == 2: b[pos:pos] = bytes(rng.randrange(32, 127) for _ in range(rng.randrange(1, 50)))
        else: b[pos:pos] = b"\n"
    p = os.path.join(d, "fuzz.ncrf")
    open(p, "wb").write(bytes(b))
    os.environ["CFH_PARSE_MIN_SLICE"] = str(rng.choice([1, 300, 1 << 20]))     # slice seams of the threaded scan at odd places
    try:
        pk = _host.parse_report(p, min_record_len=rng.choice([1, 1000, 5000]), n_threads=rng.choice([1, 3, 8]))
        pk.units(1); pk.classify(50000)
        ok += 1
    except _host.HostError:
        bad += 1
print("FUZZ", ok, bad)
# the binary pack cache: damaged cache files are refused (checksum, end marker, shapes) and the report is parsed again
cache = os.path.join(d, "cache")
a = _host.parse_report(src, keep_rows=True, cache_dir=cache)
fn = os.path.join(cache, os.listdir(cache)[0])
blob = open(fn, "rb").read()
hits = 0
for i in range(60):
    b = bytearray(blob)
    what = rng.randrange(3)
    if what == 0: b = b[: rng.randrange(0, len(b))]
    elif what == 1: b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
    else:
        pos = rng.randrange(len(b)); b[pos:pos + 8] = bytes(rng.randrange(256) for _ in range(8))
    open(fn, "wb").write(bytes(b))
    pk = _host.parse_report(src, keep_rows=True, cache_dir=cache)
    hits += pk.from_cache
    assert pk.ids == a.ids and bytes(pk.bases) == bytes(a.bases)
    pk.units(2); pk.classify(50000)
print("CACHE", hits)
'''


@pytest.fixture(scope="module")
def asan_env():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "centroflye_amd", "csrc"), "asan"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    return dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
                CF_HOST_LIB=ASAN_LIB)


def test_host_suite_under_sanitizers(asan_env):
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_host.py"),
                        os.path.join(ROOT, "tests", "test_polisher_export.py")], env=asan_env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "passed" in p.stdout and "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr


def test_corrupted_reports_never_trip_the_sanitizers(asan_env, tmp_path):
    p = subprocess.run([sys.executable, "-c", FUZZ % dict(root=ROOT, tmp=str(tmp_path))], env=asan_env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("FUZZ ")][0].split()
    assert int(line[1]) + int(line[2]) == 300 and int(line[2]) > 20 and int(line[1]) > 20
    assert [ln for ln in p.stdout.splitlines() if ln.startswith("CACHE ")] == ["CACHE 0"]      # no damaged cache was taken for a good one
