"""Process-wide device session shared by the drop-in modules: one Engine (one GPU) per process,
as the reference runs each stage in its own single-threaded process (centroFlye.py:157-210)."""
import os

from .engine import Engine

_engine = None
_loaded = None      # (packed, n_motif) currently resident in HBM — a strong reference: id() of a collected object is reused
_clouds_token = None


def engine():
    global _engine
    if _engine is None:
        _engine = Engine(int(os.environ.get("CF_DEVICE", "0")))   # raises without GPU / library: no fallback
    return _engine


def ensure_loaded(packed, n_motif):
    """Make sure the reads of `packed` with the n_motif unit split are resident."""
    global _loaded, _clouds_token
    if _loaded is None or _loaded[0] is not packed or _loaded[1] != int(n_motif):
        engine().load(packed, n_motif)
        _loaded = (packed, int(n_motif))
        _clouds_token = None
    return engine()


def clouds_token():
    return _clouds_token


def set_clouds_token(tok):
    global _clouds_token
    _clouds_token = tok


def reset():
    global _engine, _loaded, _clouds_token
    if _engine is not None:
        _engine.close()
    _engine = _loaded = _clouds_token = None
