"""TEST INFRASTRUCTURE — read-recruitment oracle (reference scripts/read_recruitment/rr.cpp:73-90).  Only tests/,
smoke() and the bench's cpu leg may import this.

  distance(unit, read, k)   the plain-C restatement (oracle/c/cf_oracle_rr.c: Myers / Hyyro block bit-vector, HW mode)
  ref_distance(...)         the REFERENCE's own code (vendored edlib built by oracle/ref/Makefile into oracle/_ref/),
                            None when the library is absent
  recruited(unit, reads, k) the rr.cpp decision: forward or reverse-complement distance within the threshold
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_c = None
_ref = False


def _clib():
    global _c
    if _c is None:
        L = C.CDLL(os.path.join(HERE, "c", "libcforacle.so"))
        L.cfo_rr_distance.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_longlong, C.c_int]
        L.cfo_rr_revcomp.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        _c = L
    return _c


def _reflib():
    global _ref
    if _ref is False:
        path = os.path.join(HERE, "_ref", "librr_ref.so")
        _ref = None
        if os.path.exists(path):
            L = C.CDLL(path)
            L.rr_ref_distance.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int]
            _ref = L
    return _ref


def distance(unit: bytes, read: bytes, k: int) -> int:
    return _clib().cfo_rr_distance(unit, len(unit), read, len(read), k)


def ref_distance(unit: bytes, read: bytes, k: int):
    L = _reflib()
    return None if L is None else L.rr_ref_distance(unit, len(unit), read, len(read), k)


def revcomp(unit: bytes) -> bytes:
    out = C.create_string_buffer(len(unit))
    if _clib().cfo_rr_revcomp(unit, len(unit), out) != 0:
        raise ValueError("unit has a character outside upper-case ACGT (the reference asserts, rr.cpp:24)")
    return out.raw


def recruited(unit: bytes, reads, k: int):
    rc = revcomp(unit)
    return [distance(unit, r, k) != -1 or distance(rc, r, k) != -1 for r in reads]
