"""ctypes binding of libcfhost.so (include/cfhost.h): NCRF ingestion + synthetic generator.

Host-side only; no GPU involved.  The library is built in-tree by
``centroflye_amd/csrc/Makefile`` (``__graft_entry__.build()``) and must be present — there
is no pure-Python fallback.
"""
import ctypes as C
import hashlib
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("CF_HOST_LIB") or os.path.join(_HERE, "libcfhost.so")     # CF_HOST_LIB: the sanitizer build (tests)


class SynthParams(C.Structure):
    """Mirror of ``cfh_synth_params`` (include/cfhost.h)."""
    _fields_ = [
        ("seed", C.c_uint64), ("unit_len", C.c_int32), ("monomer_len", C.c_int32),
        ("monomer_div", C.c_double), ("n_units", C.c_int64), ("flank", C.c_int64),
        ("unit_div", C.c_double), ("n_reads", C.c_int64), ("mean_len", C.c_double),
        ("sigma", C.c_double), ("min_len", C.c_int64), ("max_len", C.c_int64),
        ("p_del", C.c_double), ("p_sub", C.c_double), ("p_ins", C.c_double),
        ("min_aligned", C.c_int64), ("n_prefix", C.c_int32), ("n_suffix", C.c_int32),
        ("prefix_threshold", C.c_int64), ("p_split", C.c_double), ("n_threads", C.c_int32),
        ("var_len", C.c_int32), ("cand_offset", C.c_int32), ("cand_stride", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"{_LIB_PATH} is missing: build it with `make -C centroflye_amd/csrc host` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
    L = C.CDLL(_LIB_PATH)
    P = C.c_void_p
    i64, i32 = C.c_int64, C.c_int32
    pi64 = C.POINTER(C.c_int64)
    L.cfh_synth_defaults.argtypes = [C.POINTER(SynthParams)]
    L.cfh_synth_defaults.restype = None
    L.cfh_synth.argtypes = [C.POINTER(SynthParams), C.c_char_p, C.c_int, C.POINTER(P), C.c_char_p, C.c_int]
    L.cfh_parse_report.argtypes = [C.c_char_p, i64, C.c_int, C.c_int, C.POINTER(P), C.c_char_p, C.c_int]
    L.cfh_pack_save.argtypes = [P, C.c_char_p, C.POINTER(C.c_int64), C.c_char_p, C.c_int]
    L.cfh_pack_load.argtypes = [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(P), C.c_char_p, C.c_int]
    L.cfh_pack_free.argtypes = [P]
    L.cfh_pack_free.restype = None
    for name in ("cfh_n_reads", "cfh_n_bases", "cfh_n_seen"):
        getattr(L, name).argtypes = [P]
        getattr(L, name).restype = i64
    L.cfh_non_acgt.argtypes = [P]
    L.cfh_non_acgt.restype = i32
    L.cfh_exotic_summary.argtypes = [P, i32, i32, C.c_uint32, C.c_uint32, i64, i64, pi64]
    L.cfh_exotic_list.restype = i64
    L.cfh_exotic_list.argtypes = [P, i32, i64, i64, C.c_void_p, i64]
    L.cfh_exotic_kept.restype = i64
    L.cfh_exotic_kept.argtypes = [P, i32, i32, C.c_void_p, C.c_void_p, i64]
    L.cfh_exotic_rare.restype = i64
    L.cfh_exotic_rare.argtypes = [P, i32, i32, C.c_uint32, C.c_uint32, C.c_void_p, i64]
    L.cfh_exotic_rare_lower.restype = i64
    L.cfh_exotic_rare_lower.argtypes = [P, i32, i32, C.c_uint32, C.c_uint32, C.c_void_p, i64]
    for name in ("cfh_bases", "cfh_read_off", "cfh_ids", "cfh_id_off", "cfh_meta"):
        getattr(L, name).argtypes = [P]
        getattr(L, name).restype = C.c_void_p
    L.cfh_n_motifs.argtypes = [P]
    L.cfh_n_motifs.restype = i32
    L.cfh_motif.argtypes = [P, i32, pi64]
    L.cfh_motif.restype = C.c_void_p
    L.cfh_discarded.argtypes = [P, pi64]
    L.cfh_discarded.restype = C.c_void_p
    L.cfh_units.argtypes = [P, i32, pi64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                            C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    L.cfh_classify.argtypes = [P, i64, i64, C.c_void_p]
    L.cfh_row.argtypes = [P, i64, i32, pi64]
    L.cfh_row.restype = C.c_void_p
    L.cfh_write_kmers.argtypes = [C.c_char_p, C.c_void_p, i64, i32, C.c_char_p, C.c_int]
    L.cfh_write_edges.argtypes = [C.c_char_p, C.c_int, C.c_void_p, i32, C.c_void_p, i64, C.c_char_p, C.c_int]
    L.cfh_read_kmers.argtypes = [C.c_char_p, i32, C.c_void_p, i64, pi64, C.c_char_p, C.c_int]
    L.cfh_export_read_units.argtypes = [P, C.c_void_p, C.c_void_p, i64, i64, i64, C.c_char_p, C.c_int, pi64, pi64, C.c_char_p, C.c_int]
    _lib = L
    return L


class HostError(RuntimeError):
    pass


def _check(rc, err):
    if rc != 0:
        raise HostError(f"cfhost error {rc}: {err.value.decode(errors='replace')}")


def _view(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


class PackedReads:
    """Flat-array form of an NCRF report.

    Attributes (numpy views borrowed from the native pack; valid while this object lives):
      bases     uint8[N_b]  ASCII de-gapped oriented read rows
      read_off  int64[R+1]
      ids       list[str]   read ids in record order
      meta      int64[R,8]  r_len, r_al_len, r_st, r_en, strand(0/1), n_alignments, ncols, motif id
    """

    def __init__(self, handle):
        self._h = handle
        L = lib()
        self.n_reads = int(L.cfh_n_reads(handle))
        self.n_bases = int(L.cfh_n_bases(handle))
        self.non_acgt = bool(L.cfh_non_acgt(handle))
        R = self.n_reads
        self.bases = _view(L.cfh_bases(handle), self.n_bases, np.uint8)
        self.read_off = _view(L.cfh_read_off(handle), R + 1, np.int64)
        id_off = _view(L.cfh_id_off(handle), R + 1, np.int64)
        raw = _view(L.cfh_ids(handle), int(id_off[-1]) if R else 0, np.uint8).tobytes()
        self.ids = [raw[id_off[i]:id_off[i + 1]].decode() for i in range(R)]
        self.meta = _view(L.cfh_meta(handle), R * 8, np.int64).reshape(R, 8)
        self.motifs = []
        for m in range(L.cfh_n_motifs(handle)):
            n = C.c_int64()
            p = L.cfh_motif(handle, m, C.byref(n))
            self.motifs.append(C.string_at(p, n.value).decode())
        n = C.c_int64()
        p = L.cfh_discarded(handle, C.byref(n))
        self.discarded_reads = [x for x in C.string_at(p, n.value).decode().split("\n") if x] if n.value else []

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.cfh_pack_free(h)

    def units(self, n=1):
        """(unit_ptr[R+1], unit_start[U], unit_end[U], unit_col[U,2]) for n_motif = n."""
        L = lib()
        err = C.create_string_buffer(512)
        nu = C.c_int64()
        p = [C.c_void_p() for _ in range(4)]
        _check(L.cfh_units(self._h, n, C.byref(nu), C.byref(p[0]), C.byref(p[1]), C.byref(p[2]),
                           C.byref(p[3]), err, 512), err)
        U = nu.value
        return (_view(p[0].value, self.n_reads + 1, np.int64), _view(p[1].value, U, np.int64),
                _view(p[2].value, U, np.int64), _view(p[3].value, 2 * U, np.int64).reshape(U, 2))

    def classify(self, large_threshold, small_threshold=1000):
        """uint8[R]: 0 prefix, 1 internal, 2 suffix (reference ncrf_parser.py:120-145)."""
        out = np.zeros(self.n_reads, dtype=np.uint8)
        lib().cfh_classify(self._h, int(large_threshold), int(small_threshold), out.ctypes.data)
        return out

    def exotic_summary(self, k, max_nonuniq, lo, hi, read_lo=0, read_hi=None):
        """The k-mer windows that hold a symbol other than upper-case A, C, G, T, counted as the reference counts every
        window (scripts/distance_based_kmer_recruitment.py:39-63, strings of the raw row): dict with n_distinct, n_read_kmers,
        n_kept (multi <= max_nonuniq), n_rare (in the rare window) and n_blocking (rare AND free of lower-case letters:
        only those could match a window of an upper-cased unit, read_kmer_cloud.py:25, and reach the outputs).  The device
        path skips these windows; with n_blocking == 0 its outputs equal the reference's."""
        out = (C.c_int64 * 5)()
        rc = lib().cfh_exotic_summary(self._h, int(k), int(max_nonuniq), int(lo), int(min(hi, 2 ** 32 - 1)), int(read_lo),
                                      int(self.n_reads if read_hi is None else read_hi), out)
        if rc:
            raise HostError(f"cfh_exotic_summary failed ({rc})")
        return dict(n_distinct=out[0], n_read_kmers=out[1], n_kept=out[2], n_rare=out[3], n_blocking=out[4])

    def exotic_list(self, k, read_lo=0, read_hi=None):
        """int64[n, 5]: per distinct window with a symbol other than upper-case A, C, G, T: two hashes of its text, pres, multi,
        1 if it holds no lower-case letter — counts that ADD over disjoint read sets (one row per window and shard)."""
        hi = self.n_reads if read_hi is None else read_hi
        n = lib().cfh_exotic_list(self._h, int(k), int(read_lo), int(hi), None, 0)
        if n < 0:
            raise HostError(f"cfh_exotic_list failed ({n})")
        rows = np.zeros((n, 5), np.int64)
        if n and lib().cfh_exotic_list(self._h, int(k), int(read_lo), int(hi), rows.ctypes.data, n) != n:
            raise HostError("cfh_exotic_list: the window set changed between two calls")
        return rows

    def exotic_kept(self, k, max_nonuniq):
        """{window text: reads holding it} for every window with a symbol other than upper-case A, C, G, T that the reference's table
        keeps (twice in at most max_nonuniq reads)."""
        n = lib().cfh_exotic_kept(self._h, int(k), int(max_nonuniq), None, None, 0)
        if n < 0:
            raise HostError(f"cfh_exotic_kept failed ({n})")
        buf = C.create_string_buffer(max(1, n * int(k)))
        pres = np.zeros(max(1, n), np.int64)
        if n and lib().cfh_exotic_kept(self._h, int(k), int(max_nonuniq), buf, pres.ctypes.data, n) != n:
            raise HostError("cfh_exotic_kept: the window set changed between two calls")
        raw = buf.raw[:n * int(k)].decode("latin-1")
        return {raw[i * k:(i + 1) * k]: int(pres[i]) for i in range(n)}

    def exotic_rare_lower(self, k, max_nonuniq, lo, hi):
        """The rare windows that hold a lower-case letter, as strings in ascending order: members of the reference's rare SET only (they
        never equal a window of an upper-cased unit)."""
        args = (self._h, int(k), int(max_nonuniq), int(lo), int(min(hi, 2 ** 32 - 1)))
        n = lib().cfh_exotic_rare_lower(*args, None, 0)
        if n < 0:
            raise HostError(f"cfh_exotic_rare_lower failed ({n})")
        buf = C.create_string_buffer(max(1, n * int(k)))
        if n and lib().cfh_exotic_rare_lower(*args, buf, n) != n:
            raise HostError("cfh_exotic_rare_lower: the window set changed between two calls")
        raw = buf.raw[:n * int(k)].decode("latin-1")
        return [raw[i * k:(i + 1) * k] for i in range(n)]

    def exotic_rare(self, k, max_nonuniq, lo, hi):
        """The rare windows that hold a symbol other than A, C, G, T and no lower-case letter, as strings in ascending order
        (exotic_summary's n_blocking of them): the reference selects these k-mers like any other."""
        args = (self._h, int(k), int(max_nonuniq), int(lo), int(min(hi, 2 ** 32 - 1)))
        n = lib().cfh_exotic_rare(*args, None, 0)
        if n < 0:
            raise HostError(f"cfh_exotic_rare failed ({n})")
        buf = C.create_string_buffer(max(1, n * int(k)))
        if n and lib().cfh_exotic_rare(*args, buf, n) != n:
            raise HostError("cfh_exotic_rare: the window set changed between two calls")
        raw = buf.raw[:n * int(k)].decode("latin-1")
        return [raw[i * k:(i + 1) * k] for i in range(n)]

    def export_read_units(self, rec, pos, outdir, min_pos=0, max_pos=None, n_threads=0):
        """Per-position read-unit FASTA files (reference eltr_polisher.py:53-97).  rec / pos: record indices and
        positions of the placed reads in read_positions.csv order; max_pos None = infinity.  Returns
        (number of positions, number of units written)."""
        rec = np.ascontiguousarray(rec, dtype=np.int64)
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        err = C.create_string_buffer(512)
        n_pos, n_units = C.c_int64(), C.c_int64()
        _check(lib().cfh_export_read_units(self._h, rec.ctypes.data, pos.ctypes.data, rec.size, int(min_pos),
                                           -1 if max_pos is None else int(max_pos), os.fsencode(outdir), int(n_threads),
                                           C.byref(n_pos), C.byref(n_units), err, 512), err)
        return n_pos.value, n_units.value

    def row(self, r, which):
        n = C.c_int64()
        p = lib().cfh_row(self._h, r, which, C.byref(n))
        if not p:
            raise HostError("alignment rows were not kept (parse with keep_rows=True)")
        return C.string_at(p, n.value).decode()


def parse_report(path, min_record_len=5000, keep_rows=True, n_threads=0, cache_dir=None):
    """Parse an NCRF report into packed arrays.  cache_dir (default: $CF_PACK_CACHE, unset = no cache): directory of binary
    packs keyed by the report's path, size, mtime and the two options — the second stage script that reads the same report
    loads the arrays instead of parsing the text again."""
    L = lib()
    err = C.create_string_buffer(512)
    h = C.c_void_p()
    cache_dir = os.environ.get("CF_PACK_CACHE") if cache_dir is None else cache_dir
    cache_fn = src = None
    if cache_dir:
        try:
            st = os.stat(path)
            src = (C.c_int64 * 4)(st.st_size, st.st_mtime_ns, int(min_record_len), int(bool(keep_rows)))
            key = hashlib.sha1(f"{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}|{int(min_record_len)}|{int(bool(keep_rows))}".encode()).hexdigest()
            cache_fn = os.path.join(cache_dir, key + ".cfpack")
            if os.path.exists(cache_fn) and L.cfh_pack_load(os.fsencode(cache_fn), src, C.byref(h), err, 512) == 0:
                pk = PackedReads(h)
                pk.from_cache = True
                return pk
        except OSError:
            cache_fn = None
    _check(L.cfh_parse_report(os.fsencode(path), min_record_len, int(keep_rows), n_threads,
                              C.byref(h), err, 512), err)
    pk = PackedReads(h)
    pk.from_cache = False
    if cache_fn:
        try:
            os.makedirs(cache_dir, exist_ok=True)
            L.cfh_pack_save(h, os.fsencode(cache_fn), src, err, 512)      # best effort: a cache that cannot be written is no error
        except OSError:
            pass
    return pk


def synth_params(**kw):
    sp = SynthParams()
    lib().cfh_synth_defaults(C.byref(sp))
    for k, v in kw.items():
        if not hasattr(sp, k):
            raise TypeError(f"unknown synth parameter {k}")
        setattr(sp, k, v)
    return sp


def synth(report_path=None, pack=True, keep_rows=False, **kw):
    """Run the synthetic generator; returns PackedReads (or None when pack=False)."""
    L = lib()
    sp = synth_params(**kw)
    err = C.create_string_buffer(512)
    h = C.c_void_p()
    _check(L.cfh_synth(C.byref(sp), os.fsencode(report_path) if report_path else None,
                       int(keep_rows), C.byref(h) if pack else None, err, 512), err)
    return PackedReads(h) if pack else None


def write_kmers(path, kmers, k):
    kmers = np.ascontiguousarray(kmers, dtype=np.uint64)
    err = C.create_string_buffer(512)
    _check(lib().cfh_write_kmers(os.fsencode(path), kmers.ctypes.data, kmers.size, k, err, 512), err)


def write_edges(path, rare_kmers, k, edges, append=False):
    rare_kmers = np.ascontiguousarray(rare_kmers, dtype=np.uint64)
    edges = np.ascontiguousarray(edges, dtype=np.uint32).reshape(-1, 4)
    err = C.create_string_buffer(512)
    _check(lib().cfh_write_edges(os.fsencode(path), int(append), rare_kmers.ctypes.data, k,
                                 edges.ctypes.data, edges.shape[0], err, 512), err)


def read_kmers(path, k):
    L = lib()
    err = C.create_string_buffer(512)
    n = C.c_int64()
    _check(L.cfh_read_kmers(os.fsencode(path), k, None, 0, C.byref(n), err, 512), err)
    out = np.zeros(n.value, dtype=np.uint64)
    _check(L.cfh_read_kmers(os.fsencode(path), k, out.ctypes.data, out.size, C.byref(n), err, 512), err)
    return out
