"""Flat-import shim: the reference's scripts import their siblings as `from cloud_contig import ...`."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centroflye_amd.cloud_contig import *  # noqa: E402,F401,F403
