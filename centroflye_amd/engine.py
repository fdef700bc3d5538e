"""Thin numpy-facing wrapper over the C ABI of the device pipeline (include/cfhip.h).

One ``Engine`` = one ``cf_ctx`` = one GPU.  Every method maps 1:1 onto a C entry point; the
mirrors of the reference's modules (``distance_based_kmer_recruitment``, ``read_kmer_cloud``,
``cloud_contig``, ``read_placer``) are written on top of this class.
"""
import ctypes as C

import numpy as np

from . import _lib


class DeviceError(RuntimeError):
    pass


def _ptr(a):
    return a.ctypes.data if a is not None else None


class Engine:
    def __init__(self, device=0, lib=None):
        self._lib = lib if lib is not None else _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.cf_create(int(device), C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.cf_last_error(self._ctx).decode(errors="replace") if self._ctx else "allocation failed"
            if self._ctx:
                self._lib.cf_destroy(self._ctx)
                self._ctx = C.c_void_p()
            raise DeviceError(f"cf_create({device}) failed ({rc}): {msg}")
        self.k = None
        self.n_reads = 0
        self.n_units = 0

    # ------------------------------------------------------------------ plumbing
    def close(self):
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx:
            self._lib.cf_destroy(ctx)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise DeviceError(f"{what} failed ({rc}): {self._lib.cf_last_error(self._ctx).decode(errors='replace')}")

    def device_info(self):
        name = C.create_string_buffer(256)
        hbm, cu = C.c_int64(), C.c_int32()
        self._check(self._lib.cf_device_info(self._ctx, name, 256, C.byref(hbm), C.byref(cu)), "cf_device_info")
        return dict(name=name.value.decode(), hbm_bytes=hbm.value, n_cu=cu.value)

    def set_param(self, name, value):
        self._check(self._lib.cf_set_param(self._ctx, name.encode(), int(value)), f"cf_set_param({name})")

    def stats(self):
        s = _lib.Stats()
        self._check(self._lib.cf_get_stats(self._ctx, C.byref(s)), "cf_get_stats")
        return s.as_dict()

    def times(self):
        t = _lib.Times()
        self._check(self._lib.cf_get_times(self._ctx, C.byref(t)), "cf_get_times")
        return t.as_dict()

    # ------------------------------------------------------------------ A0 hand-over
    def load_arrays(self, bases, read_off, unit_ptr, unit_start, unit_end):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        unit_ptr = np.ascontiguousarray(unit_ptr, dtype=np.int64)
        unit_start = np.ascontiguousarray(unit_start, dtype=np.int64)
        unit_end = np.ascontiguousarray(unit_end, dtype=np.int64)
        R = read_off.size - 1
        if unit_ptr.size != R + 1 or unit_start.size != unit_end.size or (R >= 0 and unit_start.size != unit_ptr[-1]):
            raise ValueError("inconsistent read / unit arrays")
        self._check(self._lib.cf_load_reads(self._ctx, _ptr(bases), _ptr(read_off), R, _ptr(unit_ptr),
                                            _ptr(unit_start), _ptr(unit_end)), "cf_load_reads")
        self.n_reads, self.n_units = R, int(unit_start.size)

    def load(self, packed, n_motif=1):
        """packed: centroflye_amd._host.PackedReads."""
        unit_ptr, unit_start, unit_end, _ = packed.units(n_motif)
        self.load_arrays(packed.bases, packed.read_off, unit_ptr, unit_start, unit_end)

    def load_units(self, unit_ptr, unit_start, unit_end):
        unit_ptr = np.ascontiguousarray(unit_ptr, dtype=np.int64)
        unit_start = np.ascontiguousarray(unit_start, dtype=np.int64)
        unit_end = np.ascontiguousarray(unit_end, dtype=np.int64)
        self._check(self._lib.cf_load_units(self._ctx, _ptr(unit_ptr), _ptr(unit_start), _ptr(unit_end)), "cf_load_units")
        self.n_units = int(unit_start.size)

    # ------------------------------------------------------------------ A1 / A2
    def count_kmers(self, k, read_lo=0, read_hi=None):
        self._check(self._lib.cf_count_kmers(self._ctx, int(k), int(read_lo),
                                             int(self.n_reads if read_hi is None else read_hi)), "cf_count_kmers")
        self.k = int(k)

    def table(self, sort=True):
        """(keys uint64, pres uint32, multi uint32) of every k-mer seen."""
        n = C.c_int64()
        self._check(self._lib.cf_get_table(self._ctx, None, None, None, 0, C.byref(n)), "cf_get_table")
        keys = np.zeros(n.value, np.uint64)
        pres = np.zeros(n.value, np.uint32)
        multi = np.zeros(n.value, np.uint32)
        if n.value:
            self._check(self._lib.cf_get_table(self._ctx, _ptr(keys), _ptr(pres), _ptr(multi), n.value, C.byref(n)), "cf_get_table")
        if sort:
            o = np.argsort(keys, kind="stable")
            keys, pres, multi = keys[o], pres[o], multi[o]
        return keys, pres, multi

    def count_occurrences(self, k, read_lo=0, read_hi=None):
        """Total occurrence counts of every k-mer; table() then returns count = pres | multi << 32."""
        self._check(self._lib.cf_count_occurrences(self._ctx, int(k), int(read_lo),
                                                   int(self.n_reads if read_hi is None else read_hi)), "cf_count_occurrences")
        self.k = int(k)

    def top_kmers(self, n):
        """(keys uint64, counts uint64) of the n k-mers with the largest (count, k-mer), descending."""
        m = C.c_int64()
        self._check(self._lib.cf_top_kmers(self._ctx, int(n), None, None, C.byref(m)), "cf_top_kmers")
        keys = np.zeros(m.value, np.uint64)
        counts = np.zeros(m.value, np.uint64)
        if m.value:
            self._check(self._lib.cf_top_kmers(self._ctx, int(n), _ptr(keys), _ptr(counts), C.byref(m)), "cf_top_kmers")
        return keys, counts

    def rr_distances(self, unit, reads, read_off, threshold):
        """(dist_fwd, dist_rc) int32[n_reads]: minimum edit distance between the unit / its reverse complement and a
        substring of each read, -1 above the threshold (reference rr.cpp:73-90).  reads: uint8 bytes back to back."""
        unit = np.frombuffer(bytes(unit), dtype=np.uint8)
        reads = np.ascontiguousarray(reads, dtype=np.uint8)
        read_off = np.ascontiguousarray(read_off, dtype=np.int64)
        n = read_off.size - 1
        fwd = np.zeros(max(n, 0), np.int32)
        rc = np.zeros(max(n, 0), np.int32)
        self._check(self._lib.cf_rr_distances(self._ctx, _ptr(unit), int(unit.size), _ptr(reads) if reads.size else None, _ptr(read_off),
                                              int(n), int(threshold), _ptr(fwd) if n else None, _ptr(rc) if n else None), "cf_rr_distances")
        return fwd, rc

    def reset_table(self, k, expected_keys):
        self._check(self._lib.cf_reset_table(self._ctx, int(k), int(expected_keys)), "cf_reset_table")
        self.k = int(k)

    def table_size(self):
        n = C.c_int64()
        self._check(self._lib.cf_get_table(self._ctx, None, None, None, 0, C.byref(n)), "cf_get_table")
        return n.value

    # raw-pointer variants: the buffers may live on the host or on this GPU (e.g. torch tensors used
    # for the RCCL exchange); the library copies with hipMemcpyDefault
    def table_into(self, keys_ptr, pres_ptr, multi_ptr, cap):
        n = C.c_int64()
        self._check(self._lib.cf_get_table(self._ctx, keys_ptr, pres_ptr, multi_ptr, int(cap), C.byref(n)), "cf_get_table")
        return n.value

    def merge_table_ptr(self, keys_ptr, pres_ptr, multi_ptr, n):
        self._check(self._lib.cf_merge_table(self._ctx, keys_ptr, pres_ptr, multi_ptr, int(n)), "cf_merge_table")

    def kmers_into(self, ptr, cap):
        self._check(self._lib.cf_get_kmers(self._ctx, ptr, int(cap)), "cf_get_kmers")

    def set_kmers_ptr(self, ptr, n, k):
        self._check(self._lib.cf_set_kmers(self._ctx, ptr, int(n), int(k)), "cf_set_kmers")
        self.k = int(k)

    def clouds_into(self, cloud_ptr_ptr, entries_ptr, cap):
        self._check(self._lib.cf_get_clouds(self._ctx, cloud_ptr_ptr, entries_ptr, int(cap)), "cf_get_clouds")

    def set_clouds_ptr(self, cloud_ptr_ptr, entries_ptr, n_entries):
        self._check(self._lib.cf_set_clouds(self._ctx, cloud_ptr_ptr, entries_ptr, int(n_entries)), "cf_set_clouds")

    def unique_mask_into(self, ptr):
        self._check(self._lib.cf_get_unique_mask(self._ctx, ptr), "cf_get_unique_mask")

    def or_unique_mask_ptr(self, ptr):
        self._check(self._lib.cf_or_unique_mask(self._ctx, ptr), "cf_or_unique_mask")

    def merge_table(self, keys, pres, multi):
        keys = np.ascontiguousarray(keys, np.uint64)
        pres = np.ascontiguousarray(pres, np.uint32)
        multi = np.ascontiguousarray(multi, np.uint32)
        self._check(self._lib.cf_merge_table(self._ctx, _ptr(keys), _ptr(pres), _ptr(multi), keys.size), "cf_merge_table")

    def select_rare(self, max_nonuniq, lo, hi):
        n = C.c_int64()
        lo, hi = max(0, int(lo)), int(hi)
        if hi < lo or hi < 0:
            lo, hi = 1, 0
        self._check(self._lib.cf_select_rare(self._ctx, int(max_nonuniq), lo, min(hi, 2 ** 32 - 1), C.byref(n)), "cf_select_rare")
        return n.value

    def set_kmers(self, kmers, k):
        kmers = np.ascontiguousarray(kmers, np.uint64)
        self._check(self._lib.cf_set_kmers(self._ctx, _ptr(kmers), kmers.size, int(k)), "cf_set_kmers")
        self.k = int(k)

    def kmers(self):
        n = self.stats()["n_kmers"]
        out = np.zeros(n, np.uint64)
        self._check(self._lib.cf_get_kmers(self._ctx, _ptr(out), n), "cf_get_kmers")
        return out

    # ------------------------------------------------------------------ A3 / A4
    def build_clouds(self):
        n = C.c_int64()
        self._check(self._lib.cf_build_clouds(self._ctx, C.byref(n)), "cf_build_clouds")
        return n.value

    def filter_clouds(self, min_mult=2, max_mult=0):
        n = C.c_int64()
        self._check(self._lib.cf_filter_clouds(self._ctx, int(min_mult), int(max_mult), C.byref(n)), "cf_filter_clouds")
        return n.value

    def clouds(self):
        """(cloud_ptr int64[U+1], entries int32[N_ce])."""
        n = self.stats()["n_cloud_entries"]
        ptr = np.zeros(self.n_units + 1, np.int64)
        ent = np.zeros(n, np.int32)
        self._check(self._lib.cf_get_clouds(self._ctx, _ptr(ptr), _ptr(ent), n), "cf_get_clouds")
        return ptr, ent

    def set_clouds(self, cloud_ptr, entries):
        cloud_ptr = np.ascontiguousarray(cloud_ptr, np.int64)
        entries = np.ascontiguousarray(entries, np.int32)
        self._check(self._lib.cf_set_clouds(self._ctx, _ptr(cloud_ptr), _ptr(entries), entries.size), "cf_set_clouds")

    # ------------------------------------------------------------------ A5 + A6
    def dist_edges(self, min_n=0, max_n=2 ** 62, min_d=1, max_d=150, min_cov=4, rel_threshold=0.8,
                   part=0, n_parts=1, edge_cap=None):
        n = C.c_int64()
        if edge_cap is None:
            edge_cap = 0
        self._check(self._lib.cf_dist_edges(self._ctx, int(min_n), int(min(max_n, 2 ** 62)), int(min_d), int(max_d),
                                            int(min_cov), float(rel_threshold), int(part), int(n_parts), int(edge_cap),
                                            C.byref(n)), "cf_dist_edges")
        return n.value

    def edges(self, n):
        """The first n stored edges as (n, 4) uint32 rows (d, a, b, cnt)."""
        out = np.zeros((n, 4), np.uint32)
        self._check(self._lib.cf_get_edges(self._ctx, _ptr(out), n), "cf_get_edges")
        return out

    def sort_edges(self):
        """Sort the stored edges by (d, a, b) on the device."""
        self._check(self._lib.cf_sort_edges(self._ctx), "cf_sort_edges")

    def edges_checksum(self, n=2 ** 62):
        """Order-independent checksum of the first n stored edges, computed on the device (see cfhip.h)."""
        out = C.c_uint64()
        self._check(self._lib.cf_edges_checksum(self._ctx, int(n), C.byref(out)), "cf_edges_checksum")
        return int(out.value)

    def checksum(self, what):
        """(sum, items) of the order-independent device-side checksum of "table" (A1), "kmers" (the installed set), "clouds"
        (the CSR) or "unique" (the k-mers whose unique bit is set) — the oracle's figures (cfhip.h: cf_checksum)."""
        out, n = C.c_uint64(), C.c_int64()
        sel = {"table": 0, "kmers": 1, "clouds": 2, "unique": 3}[what]
        self._check(self._lib.cf_checksum(self._ctx, sel, C.byref(out), C.byref(n)), "cf_checksum")
        return int(out.value), int(n.value)

    def unique_mask(self):
        n = self.stats()["n_kmers"]
        out = np.zeros(n, np.uint8)
        self._check(self._lib.cf_get_unique_mask(self._ctx, _ptr(out)), "cf_get_unique_mask")
        return out.astype(bool)

    def or_unique_mask(self, mask):
        mask = np.ascontiguousarray(mask, np.uint8)
        self._check(self._lib.cf_or_unique_mask(self._ctx, _ptr(mask)), "cf_or_unique_mask")

    def reset_unique(self):
        self._check(self._lib.cf_reset_unique(self._ctx), "cf_reset_unique")

    # ------------------------------------------------------------------ multi-GPU
    def comm_init(self, rank, world, rendezvous=None):
        self._check(self._lib.cf_comm_init(self._ctx, int(rank), int(world), (rendezvous or "").encode()), "cf_comm_init")

    def comm_free(self):
        self._check(self._lib.cf_comm_free(self._ctx), "cf_comm_free")

    def comm_info(self):
        r, w = C.c_int32(), C.c_int32()
        self._check(self._lib.cf_comm_info(self._ctx, C.byref(r), C.byref(w)), "cf_comm_info")
        return r.value, w.value

    def allreduce(self, values, op="sum"):
        v = np.ascontiguousarray(values, np.int64).copy()
        self._check(self._lib.cf_comm_allreduce_i64(self._ctx, _ptr(v), v.size, 1 if op == "max" else 0), "cf_comm_allreduce_i64")
        return v

    def exchange_table(self):
        n = C.c_int64()
        self._check(self._lib.cf_exchange_table(self._ctx, C.byref(n)), "cf_exchange_table")
        return n.value

    def allgather_kmers(self):
        n = C.c_int64()
        self._check(self._lib.cf_allgather_kmers(self._ctx, C.byref(n)), "cf_allgather_kmers")
        return n.value

    def allgather_clouds(self):
        n = C.c_int64()
        self._check(self._lib.cf_allgather_clouds(self._ctx, C.byref(n)), "cf_allgather_clouds")
        return n.value

    def allreduce_unique(self):
        n = C.c_int64()
        self._check(self._lib.cf_allreduce_unique(self._ctx, C.byref(n)), "cf_allreduce_unique")
        return n.value

    # ------------------------------------------------------------------ A8 + A9
    def place_reads(self, classes, id_rank, min_cloud_kmer_freq=2, min_unit=2, min_inters=10, min_prop=3):
        """Returns (read, pos, s0, s1) arrays in the order the reference writes read_positions.csv;
        pos = -1 means None; s0 = -1 marks a prefix read (line 'r_id 0')."""
        R = self.n_reads
        classes = np.ascontiguousarray(classes, np.uint8)
        id_rank = np.ascontiguousarray(id_rank, np.int32)
        out_read = np.zeros(R, np.int64)
        out_pos = np.zeros(R, np.int64)
        out_s0 = np.zeros(R, np.int32)
        out_s1 = np.zeros(R, np.int32)
        self._check(self._lib.cf_place_reads(self._ctx, _ptr(classes), _ptr(id_rank), int(min_cloud_kmer_freq), int(min_unit),
                                             int(min_inters), int(min_prop), _ptr(out_read), _ptr(out_pos), _ptr(out_s0),
                                             _ptr(out_s1)), "cf_place_reads")
        return out_read, out_pos, out_s0, out_s1

    # ------------------------------------------------------------------ self tests of primitives
    def selftest_sort(self, keys, bits=64):
        keys = np.ascontiguousarray(keys, np.uint64)
        out = np.zeros_like(keys)
        self._check(self._lib.cf_selftest_sort(self._ctx, _ptr(keys), keys.size, int(bits), _ptr(out)), "cf_selftest_sort")
        return out

    def selftest_argmax(self, cands):
        """Winner (s0, s1, offset, rank, index, valid) of the placement's candidate reduction over rows (s0, s1, offset, rank, valid)."""
        cands = np.ascontiguousarray(cands, np.uint32).reshape(-1, 5)
        out = np.zeros(6, np.uint32)
        self._check(self._lib.cf_selftest_argmax(self._ctx, _ptr(cands) if cands.size else None, cands.shape[0], _ptr(out)), "cf_selftest_argmax")
        return out

    def selftest_scan(self, vals):
        vals = np.ascontiguousarray(vals, np.int64)
        out = np.zeros(vals.size + 1, np.int64)
        self._check(self._lib.cf_selftest_scan(self._ctx, _ptr(vals), vals.size, _ptr(out)), "cf_selftest_scan")
        return out
