#!/usr/bin/env python3
"""Entry point with the reference's script name and CLI (centroFlye.py:149, :211-222 call
`python -u scripts/eltr_polisher.py ...`); the implementation lives in centroflye_amd/eltr_polisher.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from centroflye_amd.eltr_polisher import *  # noqa: E402,F401,F403
from centroflye_amd.eltr_polisher import main  # noqa: E402

if __name__ == "__main__":
    main()
