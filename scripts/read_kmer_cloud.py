"""Flat-import shim: the reference's scripts import their siblings as `from read_kmer_cloud import ...`."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centroflye_amd.read_kmer_cloud import *  # noqa: E402,F401,F403
