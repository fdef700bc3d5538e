"""ctypes wrapper of oracle/c/cf_oracle.c (plain-C stage 2).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_SO = os.path.join(_DIR, "libcforacle.so")


class Result(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_bases", "n_windows", "n_read_kmers", "n_distinct", "n_kept", "n_rare", "n_units",
        "n_cloud_entries", "n_emissions", "n_edges", "n_unique")] + [
        ("edge_checksum", C.c_uint64), ("rare_checksum", C.c_uint64), ("cloud_checksum", C.c_uint64)]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


_lib = None


def build():
    src = os.path.join(_DIR, "cf_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _DIR])
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        P = C.c_void_p
        L.cfo_stage2.argtypes = [P, P, C.c_int64, P, P, P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int64, C.c_int64,
                                 C.c_int, C.c_int, C.c_uint32, C.c_double, C.POINTER(Result), P, C.c_int64, P, P, C.c_int64,
                                 P, C.c_int64, P]
        for f in ("cfo_edge_mix", "cfo_cloud_mix", "cfo_key_mix"):
            getattr(L, f).restype = C.c_uint64
        L.cfo_edge_mix.argtypes = [C.c_uint64] * 4
        L.cfo_cloud_mix.argtypes = [C.c_uint64] * 2
        L.cfo_key_mix.argtypes = [C.c_uint64]
        _lib = L
    return _lib


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x


def edge_checksum(edges):
    """Order-independent checksum of an (n, 4) array of (d, a, b, cnt) — same mix as cf_oracle.c."""
    e = np.asarray(edges).astype(np.uint64).reshape(-1, 4)
    with np.errstate(over="ignore"):
        h = _mix64(_mix64(_mix64(_mix64(e[:, 0] + np.uint64(0x9E37)) ^ e[:, 1]) ^ (e[:, 2] << np.uint64(1))) ^ (e[:, 3] << np.uint64(2)))
        return int(h.sum(dtype=np.uint64))


def rare_checksum(kmers):
    with np.errstate(over="ignore"):
        return int(_mix64(np.asarray(kmers, np.uint64) ^ np.uint64(0xABCDEF)).sum(dtype=np.uint64))


def cloud_checksum(cloud_ptr, entries):
    cloud_ptr = np.asarray(cloud_ptr, np.int64)
    units = np.repeat(np.arange(cloud_ptr.size - 1, dtype=np.uint64), np.diff(cloud_ptr))
    with np.errstate(over="ignore"):
        return int(_mix64(_mix64(units + np.uint64(0x51ED)) ^ np.asarray(entries).astype(np.uint64)).sum(dtype=np.uint64))


def stage2(bases, read_off, unit_ptr, unit_start, unit_end, k=19, max_nonuniq=3, lo=10, hi=32, min_n=0, max_n=2 ** 62,
           min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, want_arrays=False, edges_cap=0):
    """Run the plain-C stage 2.  Returns (counters dict, arrays dict or None)."""
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    unit_ptr = np.ascontiguousarray(unit_ptr, np.int64)
    unit_start = np.ascontiguousarray(unit_start, np.int64)
    unit_end = np.ascontiguousarray(unit_end, np.int64)
    R = read_off.size - 1
    res = Result()
    arrays = None
    args = [bases.ctypes.data, read_off.ctypes.data, R, unit_ptr.ctypes.data, unit_start.ctypes.data, unit_end.ctypes.data,
            k, max_nonuniq, lo, hi, min_n, min(max_n, 2 ** 62), min_d, max_d, min_cov, rel_threshold, C.byref(res)]
    if want_arrays:
        # two calls: sizes first, then arrays
        rc = lib().cfo_stage2(*args, None, 0, None, None, 0, None, 0, None)
        if rc:
            raise RuntimeError(f"cfo_stage2 failed ({rc})")
        rare = np.zeros(res.n_rare, np.uint64)
        cptr = np.zeros(res.n_units + 1, np.int64)
        ent = np.zeros(res.n_cloud_entries, np.int32)
        ne = res.n_edges
        edges = np.zeros((ne, 4), np.uint32)
        uniq = np.zeros(res.n_rare, np.uint8)
        rc = lib().cfo_stage2(*args, rare.ctypes.data, rare.size, cptr.ctypes.data, ent.ctypes.data, ent.size,
                              edges.ctypes.data, ne, uniq.ctypes.data)
        arrays = dict(rare=rare, cloud_ptr=cptr, entries=ent, edges=edges, unique=uniq.astype(bool))
    else:
        rc = lib().cfo_stage2(*args, None, 0, None, None, 0, None, 0, None)
    if rc:
        raise RuntimeError(f"cfo_stage2 failed ({rc})")
    return res.as_dict(), arrays
