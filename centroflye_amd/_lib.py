"""ctypes binding of libcfhip.so (include/cfhip.h): the gfx950 device pipeline.

There is NO CPU fallback: if the library is missing or no HIP device is visible, creating a
context raises.  ``load(path)`` exists so that the CPU test-suite can bind the same prototypes
to the host-emulated build of the same kernel sources (tests/emu); the package itself only
ever loads the in-tree ``libcfhip.so``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcfhip.so")


class Stats(C.Structure):
    """Mirror of ``cf_stats``."""
    _fields_ = [(n, C.c_int64) for n in (
        "n_reads", "n_bases", "n_units", "n_windows", "n_read_kmers", "n_distinct", "n_kept",
        "n_kmers", "n_cloud_entries", "n_emissions", "n_edges", "n_unique", "table_capacity",
        "n_spilled", "hbm_bytes_live", "n_dist_passes", "n_edges_stored")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Times(C.Structure):
    """Mirror of ``cf_times``."""
    _fields_ = [(n, C.c_float) for n in (
        "load_ms", "count_ms", "select_ms", "clouds_ms", "filter_ms", "postings_ms", "dist_ms",
        "place_ms", "dist_kernel_ms", "count_kernel_ms", "rr_kernel_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


# every symbol include/cfhip.h declares: name -> (restype, argtypes)
_P, _I64, _I32, _U32 = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32
_PI64 = C.POINTER(C.c_int64)
PROTOTYPES = {
    "cf_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "cf_destroy": (None, [_P]),
    "cf_last_error": (C.c_char_p, [_P]),
    "cf_device_info": (C.c_int, [_P, C.c_char_p, C.c_int, _PI64, C.POINTER(_I32)]),
    "cf_load_reads": (C.c_int, [_P, _P, _P, _I64, _P, _P, _P]),
    "cf_rr_distances": (C.c_int, [_P, _P, _I32, _P, _P, _I64, _I32, _P, _P]),
    "cf_load_units": (C.c_int, [_P, _P, _P, _P]),
    "cf_count_kmers": (C.c_int, [_P, _I32, _I64, _I64]),
    "cf_count_occurrences": (C.c_int, [_P, _I32, _I64, _I64]),
    "cf_top_kmers": (C.c_int, [_P, _I64, _P, _P, _PI64]),
    "cf_reset_table": (C.c_int, [_P, _I32, _I64]),
    "cf_get_table": (C.c_int, [_P, _P, _P, _P, _I64, _PI64]),
    "cf_merge_table": (C.c_int, [_P, _P, _P, _P, _I64]),
    "cf_select_rare": (C.c_int, [_P, _I32, _U32, _U32, _PI64]),
    "cf_set_kmers": (C.c_int, [_P, _P, _I64, _I32]),
    "cf_get_kmers": (C.c_int, [_P, _P, _I64]),
    "cf_build_clouds": (C.c_int, [_P, _PI64]),
    "cf_filter_clouds": (C.c_int, [_P, _U32, _U32, _PI64]),
    "cf_get_clouds": (C.c_int, [_P, _P, _P, _I64]),
    "cf_set_clouds": (C.c_int, [_P, _P, _P, _I64]),
    "cf_dist_edges": (C.c_int, [_P, _I64, _I64, _I32, _I32, _U32, C.c_double, _I32, _I32, _I64, _PI64]),
    "cf_get_edges": (C.c_int, [_P, _P, _I64]),
    "cf_sort_edges": (C.c_int, [_P]),
    "cf_edges_checksum": (C.c_int, [_P, _I64, C.POINTER(C.c_uint64)]),
    "cf_checksum": (C.c_int, [_P, _I32, C.POINTER(C.c_uint64), _PI64]),
    "cf_get_unique_mask": (C.c_int, [_P, _P]),
    "cf_or_unique_mask": (C.c_int, [_P, _P]),
    "cf_reset_unique": (C.c_int, [_P]),
    "cf_place_reads": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "cf_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "cf_get_times": (C.c_int, [_P, C.POINTER(Times)]),
    "cf_set_param": (C.c_int, [_P, C.c_char_p, _I64]),
    "cf_comm_init": (C.c_int, [_P, _I32, _I32, C.c_char_p]),
    "cf_comm_free": (C.c_int, [_P]),
    "cf_comm_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32)]),
    "cf_comm_allreduce_i64": (C.c_int, [_P, _P, _I64, _I32]),
    "cf_exchange_table": (C.c_int, [_P, _PI64]),
    "cf_allgather_kmers": (C.c_int, [_P, _PI64]),
    "cf_allgather_clouds": (C.c_int, [_P, _PI64]),
    "cf_allreduce_unique": (C.c_int, [_P, _PI64]),
    "cf_selftest_sort": (C.c_int, [_P, _P, _I64, _I32, _P]),
    "cf_selftest_scan": (C.c_int, [_P, _P, _I64, _P]),
    "cf_selftest_argmax": (C.c_int, [_P, _P, _I64, _P]),
}

_cache = {}


def load(path=None):
    """Load a build of the device library and bind every prototype (fails loudly)."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension must be built (`make -C centroflye_amd/csrc hip` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
