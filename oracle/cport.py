"""ctypes wrapper of oracle/c (plain-C stage 2: cf_oracle.c single thread, cf_oracle_mt.c OpenMP; plain-C placement:
cf_oracle_place.c).  TEST INFRASTRUCTURE ONLY."""
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


class PartResult(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_emissions", "n_edges", "n_unique", "n_first_kmers")] + [
        ("edge_checksum", C.c_uint64), ("secs", C.c_double), ("secs_postings", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def build():
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith(".c")]
    if not os.path.exists(_SO) or any(os.path.exists(s_) and os.path.getmtime(_SO) < os.path.getmtime(s_) for s_ in srcs):
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
        L.cfo_stage2_mt.argtypes = L.cfo_stage2.argtypes + [C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.cfo_mt_prepare.restype = C.c_void_p
        L.cfo_mt_prepare.argtypes = [P, P, C.c_int64, P, P, P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                     C.POINTER(Result), C.POINTER(C.c_uint64), P, C.POINTER(C.c_int)]
        L.cfo_mt_dist_part.argtypes = [P, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int,
                                       C.POINTER(PartResult), P, P, C.c_int64]
        L.cfo_mt_free.argtypes = [P]
        L.cfo_mt_free.restype = None
        L.cfo_mt_get.argtypes = [P, P, P, P]
        L.cfo_place_reads.argtypes = [C.c_int64, C.c_int64, P, P, P, P, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, P]
        L.cfo_table_mix.restype = C.c_uint64
        L.cfo_table_mix.argtypes = [C.c_uint64] * 3
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


def table_checksum(keys, pres, multi):
    """Order-independent checksum of a (key, pres, multi) table — same mix as cf_oracle_mt.c (cfo_table_mix)."""
    with np.errstate(over="ignore"):
        h = _mix64(_mix64(_mix64(np.asarray(keys, np.uint64) + np.uint64(0x7AB1E)) ^ np.asarray(pres).astype(np.uint64))
                   ^ (np.asarray(multi).astype(np.uint64) << np.uint64(1)))
        return int(h.sum(dtype=np.uint64))


def place_reads(classes, id_rank, unit_ptr, cloud_ptr, entries, n_kmers, min_cloud_kmer_freq=2, min_unit=2, min_inters=10, min_prop=3):
    """Plain-C greedy placement (cf_oracle_place.c); returns (read, pos, s0, s1) like Engine.place_reads."""
    classes = np.ascontiguousarray(classes, np.uint8)
    id_rank = np.ascontiguousarray(id_rank, np.int32)
    unit_ptr = np.ascontiguousarray(unit_ptr, np.int64)
    cloud_ptr = np.ascontiguousarray(cloud_ptr, np.int64)
    entries = np.ascontiguousarray(entries, np.int32)
    R = classes.size
    rd, pos = np.zeros(R, np.int64), np.zeros(R, np.int64)
    s0, s1 = np.zeros(R, np.int32), np.zeros(R, np.int32)
    rc = lib().cfo_place_reads(R, int(n_kmers), classes.ctypes.data, id_rank.ctypes.data, unit_ptr.ctypes.data, cloud_ptr.ctypes.data,
                               entries.ctypes.data, min_cloud_kmer_freq, min_unit, min_inters, min_prop,
                               rd.ctypes.data, pos.ctypes.data, s0.ctypes.data, s1.ctypes.data)
    if rc:
        raise RuntimeError(f"cfo_place_reads failed ({rc})")
    return rd, pos, s0, s1


def stage2(bases, read_off, unit_ptr, unit_start, unit_end, k=19, max_nonuniq=3, lo=10, hi=32, min_n=0, max_n=2 ** 62,
           min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, want_arrays=False, edges_cap=0, threads=None, stop_after=0):
    """Run the plain-C stage 2.  Returns (counters dict, arrays dict or None).  threads=None: cf_oracle.c (single thread);
    threads=N (0 = all cores): cf_oracle_mt.c (OpenMP; adds "table_checksum"; stop_after=1 ends after A2, 2 after A3)."""
    if threads is not None:
        return _stage2_mt(bases, read_off, unit_ptr, unit_start, unit_end, k, max_nonuniq, lo, hi, min_n, max_n, min_d, max_d,
                          min_cov, rel_threshold, want_arrays, threads, stop_after)
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


def _stage2_mt(bases, read_off, unit_ptr, unit_start, unit_end, k, max_nonuniq, lo, hi, min_n, max_n, min_d, max_d, min_cov,
               rel_threshold, want_arrays, threads, stop_after):
    bases = np.ascontiguousarray(bases, np.uint8)
    read_off = np.ascontiguousarray(read_off, np.int64)
    unit_ptr = np.ascontiguousarray(unit_ptr, np.int64)
    unit_start = np.ascontiguousarray(unit_start, np.int64)
    unit_end = np.ascontiguousarray(unit_end, np.int64)
    R = read_off.size - 1
    res, tchk = Result(), C.c_uint64()
    args = [bases.ctypes.data, read_off.ctypes.data, R, unit_ptr.ctypes.data, unit_start.ctypes.data, unit_end.ctypes.data,
            k, max_nonuniq, lo, hi, min_n, min(max_n, 2 ** 62), min_d, max_d, min_cov, rel_threshold, C.byref(res)]
    tail = [int(threads), int(stop_after), C.byref(tchk)]
    arrays = None
    rc = lib().cfo_stage2_mt(*args, None, 0, None, None, 0, None, 0, None, *tail)
    if rc == 0 and want_arrays:
        rare = np.zeros(res.n_rare, np.uint64)
        cptr = np.zeros(res.n_units + 1, np.int64)
        ent = np.zeros(res.n_cloud_entries, np.int32)
        ne = res.n_edges
        edges = np.zeros((ne, 4), np.uint32)
        uniq = np.zeros(res.n_rare, np.uint8)
        rc = lib().cfo_stage2_mt(*args, rare.ctypes.data, rare.size, cptr.ctypes.data if stop_after != 1 else None,
                                 ent.ctypes.data if stop_after != 1 else None, ent.size,
                                 edges.ctypes.data if stop_after == 0 else None, ne, uniq.ctypes.data if stop_after == 0 else None, *tail)
        arrays = dict(rare=rare, cloud_ptr=cptr, entries=ent, edges=edges, unique=uniq.astype(bool))
    if rc:
        raise RuntimeError(f"cfo_stage2_mt failed ({rc})")
    out = res.as_dict()
    out["table_checksum"] = int(tchk.value)
    return out, arrays


class Stage2State:
    """A1-A3 of the OpenMP oracle run once and kept (cfo_mt_prepare); dist_part() then runs A5 + A6 for the first k-mers
    a % n_parts == part (cfo_mt_dist_part) — the checker of the 50 000-read distance stage and bench.py's same-data CPU
    baseline.  Reference: distance_based_kmer_recruitment.py:39-149 (dist_cnt[d][a] is a's own dict, :108-113)."""

    def __init__(self, bases, read_off, unit_ptr, unit_start, unit_end, k=19, max_nonuniq=3, lo=10, hi=32, threads=0):
        self._keep = [np.ascontiguousarray(bases, np.uint8), np.ascontiguousarray(read_off, np.int64), np.ascontiguousarray(unit_ptr, np.int64),
                      np.ascontiguousarray(unit_start, np.int64), np.ascontiguousarray(unit_end, np.int64)]
        b, ro, up, us, ue = self._keep
        res, tchk, rc = Result(), C.c_uint64(), C.c_int()
        secs = (C.c_double * 2)()
        self._h = lib().cfo_mt_prepare(b.ctypes.data, ro.ctypes.data, ro.size - 1, up.ctypes.data, us.ctypes.data, ue.ctypes.data, k,
                                       max_nonuniq, lo, hi, int(threads), 0, C.byref(res), C.byref(tchk), secs, C.byref(rc))
        if not self._h:
            raise RuntimeError(f"cfo_mt_prepare failed ({rc.value})")
        self.counters = res.as_dict()
        self.counters["table_checksum"] = int(tchk.value)
        self.secs_count_select, self.secs_clouds = float(secs[0]), float(secs[1])

    def arrays(self):
        c = self.counters
        rare = np.zeros(c["n_rare"], np.uint64)
        cptr = np.zeros(c["n_units"] + 1, np.int64)
        ent = np.zeros(c["n_cloud_entries"], np.int32)
        if lib().cfo_mt_get(self._h, rare.ctypes.data, cptr.ctypes.data, ent.ctypes.data):
            raise RuntimeError("cfo_mt_get failed")
        return dict(rare=rare, cloud_ptr=cptr, entries=ent)

    def dist_part(self, part, n_parts, min_n=0, max_n=2 ** 62, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8, threads=0,
                  unique=None, want_edges=0):
        """Returns the partition's counters (+ "edges" when want_edges = capacity > 0); `unique` (uint8[n_rare]) is OR-ed in place."""
        pr = PartResult()
        edges = np.zeros((want_edges, 4), np.uint32) if want_edges else None
        rc = lib().cfo_mt_dist_part(self._h, min_n, min(max_n, 2 ** 62), min_d, max_d, min_cov, rel_threshold, part, n_parts, int(threads),
                                    C.byref(pr), unique.ctypes.data if unique is not None else None,
                                    edges.ctypes.data if edges is not None else None, want_edges)
        if rc:
            raise RuntimeError(f"cfo_mt_dist_part failed ({rc})")
        out = pr.as_dict()
        if edges is not None:
            out["edges"] = edges[:out["n_edges"]]
        return out

    def close(self):
        if self._h:
            lib().cfo_mt_free(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
